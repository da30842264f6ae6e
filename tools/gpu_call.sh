set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/gputest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gputest.log; tail -3 gpurun_out/gputest.log
timeout 1200 bash tools/profile_r03_measure.sh > gpurun_out/measure.log 2>&1
for v in base fence; do DRM_HIP_LIBRARY=$PWD/tools/variants/libdrm_$v.so timeout 300 python tools/ab_rnea.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/ab_rnea.txt; done
cat gpurun_out/ab_rnea.txt
tail -30 gpurun_out/measure.log
