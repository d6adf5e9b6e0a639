set -u
export OMP_WAIT_POLICY=passive
echo "== 1000 frames, frame pairing, planes vs oracle every 250"
timeout 1500 python tests/evidence/long_run_parity.py --frames 1000 --pairing 1 --check-every 250 > gpurun_out/r04_long_run_2048_1000frames_paired.json 2> gpurun_out/r04_long_run.err; echo rc=$?
python -c "
import json; d=json.loads(open('gpurun_out/r04_long_run_2048_1000frames_paired.json').read().strip().splitlines()[-1]); print({k:d[k] for k in ('frames','pairs_integrated_in_one_sweep','planes_bit_identical_to_oracle','intermediate_mismatches','triangles','fraction_of_checked_voxels_at_max_weight','host_ms_per_async_call')})"
echo "== hunts, larger"
timeout 1200 python tests/evidence/fuzz_product_vs_oracle.py --cases 400 --seed 43 --ref-cull 0.5 > gpurun_out/r04_fuzz_product_vs_oracle_seed43.log 2>&1; echo rc=$?; tail -1 gpurun_out/r04_fuzz_product_vs_oracle_seed43.log | cut -c1-200
timeout 1200 python tests/evidence/fuzz_dropin_vs_reference.py --cases 250 --seed 44 --ref-cull 0.5 > gpurun_out/r04_fuzz_dropin_vs_reference_seed44.log 2>&1; echo rc=$?; tail -1 gpurun_out/r04_fuzz_dropin_vs_reference_seed44.log | cut -c1-200
timeout 1200 python tests/evidence/fuzz_programs.py --cases 80 --seed 45 > gpurun_out/r04_fuzz_programs_seed45.log 2>&1; echo rc=$?; tail -1 gpurun_out/r04_fuzz_programs_seed45.log | cut -c1-200
