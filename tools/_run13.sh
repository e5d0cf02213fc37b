export TMPDIR=/tmp
mkdir -p gpurun_out/r04m
echo "== default prefetch"; bash tools/ranker_iter.sh
echo "== thread prefetch"; LAPLACE_SAMPLER_PREFETCH=thread bash tools/ranker_iter.sh
echo "== thread prefetch, switch interval 2e-5"; LAPLACE_SAMPLER_SWITCH=2e-5 LAPLACE_SAMPLER_PREFETCH=thread bash tools/ranker_iter.sh
timeout -k 10 1150 python -m pytest tests/ -x -q -m gpu > gpurun_out/r04m/gpu_tests.log 2>&1; echo rc=$?; tail -n 12 gpurun_out/r04m/gpu_tests.log
