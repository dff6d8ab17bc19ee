set -u
python -m pytest tests -m gpu -x -q > gpurun_out/r4u_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r4u_tests.log
python bench.py > gpurun_out/r4u_bench.json 2> gpurun_out/r4u_bench.err; echo "bench rc=$?"
tools/collect_profiles.sh r4u
BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout -k 10 400 python bench.py --steps 10 --warmup 3 --replicas-per-gpu 16 > gpurun_out/r4u_forcedist.json 2> gpurun_out/r4u_forcedist.err; echo "forcedist rc=$?"
tools/profile_kernels.sh tools/c3_step.py r4u_c3step > gpurun_out/r4u_c3step.txt 2>&1
tools/collect_pmc_script.sh r4u_c3 tools/c3_step.py 5
timeout -k 10 200 tools/bin/mfma_sddmm_bench > gpurun_out/r4u_mfma.log 2>&1; echo "mfma rc=$?"
tools/prof_variants.sh r4u_sddmm tools/sddmm_flat_bench.py -- --variants 1:0 0:0 2:0 1:0x100 1:0x400 1:0x800 > /dev/null
tools/prof_variants.sh r4u_spmm tools/spmm_c3_bench.py -- --variants 0 16 1 129 417 > /dev/null
echo done
