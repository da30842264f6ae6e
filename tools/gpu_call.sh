cd $GRAFT_REPO_ROOT
echo prefetch; python tools/probe_robots.py 2>&1 | grep "n=" | grep "panda \|jaco\|allegro" | cut -c1-95
echo no prefetch; DRM_HIP_LIBRARY=tools/variants/libdrm_nopf.so python tools/probe_robots.py 2>&1 | grep "n=" | grep "panda \|jaco\|allegro" | cut -c1-95
