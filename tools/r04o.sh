# final re-validation after the last host-side change + the detector counter passes. gpurun --timeout 2000 -- 'bash tools/r04o.sh'
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > $O/r04o_gpu_suite.txt 2>&1; tail -3 $O/r04o_gpu_suite.txt
timeout 300 python __graft_entry__.py --smoke 2>&1 | grep -v "surya_amd.build\|amdgpu.ids" | tail -2
timeout 900 bash tools/profile_det_pmc.sh
