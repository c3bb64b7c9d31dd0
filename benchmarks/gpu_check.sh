mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_meta.py tests/test_gpu_fullsize.py -x -q -n 4 > gpurun_out/t_meta.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/t_meta.log

cd /tmp && export TMPDIR=/tmp
echo "rc=$?"
