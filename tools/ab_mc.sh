mkdir -p gpurun_out/ab_mc
for n in base mc1 mc2 mc3; do for skip in 1 0; do
  TSDF_HIP_MC_SKIP=$skip TSDF_HIP_LIB_PATH=cpu_tsdf_amd/lib/variants/$n/libtsdf_hip.so timeout 200 python bench.py --steps 45 --warmup 2 --extras 1 --scene-b 0 --cpu-baseline 0 --host-path 0 > gpurun_out/ab_mc/$n.$skip.json 2>> gpurun_out/ab_mc/err.log || echo "$n failed"
  python - "$n.$skip" gpurun_out/ab_mc/$n.$skip.json <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[2])); e=d.get('extras',{})
    print(f"{sys.argv[1]:10s} kernel_ms {d['roofline']['kernel_ms']:.3f}", e.get('reconstruct_phase_ms'), e.get('reconstruct_classify_d_bytes_requested'), e.get('reconstruct_active_cells'))
except Exception as e: print(sys.argv[1], "no result", e)
P
done; done | tee gpurun_out/ab_mc/summary.txt
