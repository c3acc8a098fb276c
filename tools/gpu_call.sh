set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02i
(timeout 1500 python -m pytest tests/test_gpu_multi.py -x -q -m gpu 2>&1 | tail -8) > gpurun_out/r02i/pytest_multi.txt
tail -4 gpurun_out/r02i/pytest_multi.txt
(MASTER_ADDR=127.0.0.1 MASTER_PORT=29561 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py --gpus 1 --force-exchange --steps 10 --warmup 3) > gpurun_out/r02i/bench_force_exchange.txt 2>&1
tail -1 gpurun_out/r02i/bench_force_exchange.txt | cut -c1-1500
(timeout 900 python tools/bench_configs.py) > gpurun_out/r02i/bench_configs.txt 2>&1
(timeout 600 python tools/bench_types.py) > gpurun_out/r02i/bench_types.txt 2>&1
(timeout 600 python tools/bench_skew.py) > gpurun_out/r02i/bench_skew.txt 2>&1
(timeout 600 tools/radix_bench --device 0 --verify --min-time 0.2) > gpurun_out/r02i/radix_bench.txt 2>&1
timeout 1800 bash tools/profile_bench.sh r02 all > gpurun_out/r02i/profile.log 2>&1
(timeout 600 python bench.py) > gpurun_out/r02i/bench.txt 2>&1
tail -1 gpurun_out/r02i/bench.txt | cut -c1-400
