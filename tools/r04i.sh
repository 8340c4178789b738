# end-of-round evidence, round 4: full GPU suite + smoke, then tools/profile_round.sh. gpurun --timeout 2400 -- 'bash tools/r04i.sh'
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > $O/r04i_gpu_suite.txt 2>&1; tail -3 $O/r04i_gpu_suite.txt
timeout 300 python __graft_entry__.py --smoke 2>&1 | grep -v "surya_amd.build\|amdgpu.ids" | tail -2
timeout 1500 bash tools/profile_round.sh r04i
