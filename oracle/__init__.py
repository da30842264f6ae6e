"""ORACLE — TEST INFRASTRUCTURE ONLY (never imported by the product package)."""
from .oracle import Oracle, build_oracle, oracle_lib_path  # noqa: F401
