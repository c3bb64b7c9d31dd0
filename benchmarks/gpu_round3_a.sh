# round 3, first GPU validation: the new full-size / launcher tests, the whole GPU suite, smoke, and the bench line
mkdir -p gpurun_out
export OTT_REQUIRE_GPU=1
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_dist_single.py -x -q -m gpu --durations=15 > gpurun_out/t_new.log 2>&1; echo "new tests rc=$?"; tail -30 gpurun_out/t_new.log
timeout 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_fullsize.py --deselect tests/test_gpu_dist_single.py > gpurun_out/t_rest.log 2>&1; echo "rest rc=$?"; tail -3 gpurun_out/t_rest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err ) 2>&1 | tail -3; echo "bench rc=$? lines=$(wc -l < gpurun_out/bench_a.json)"
cut -c1-1500 gpurun_out/bench_a.json; tail -5 gpurun_out/bench_a.err
