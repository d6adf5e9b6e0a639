#!/bin/bash
# name the colourless instance's limiter.  (a) is a thread trace possible on this image?  (b) the
# counter list of this box; (c) every SQ / SQC / TA / TCP counter that exists here over the colourless 2048^3 key and the
# headline, one small pass each (no --kernel-trace with --pmc).  Every command under its own timeout.
O=gpurun_out/r06_c01; mkdir -p $O
export TMPDIR=/tmp
ROOT=$(pwd)
cd /tmp
# (a) thread trace: needs the rocprof-trace-decoder library, which this image does not ship (find / -name '*trace*decoder*' = headers only)
timeout 300 rocprofv3 --att --att-target-cu 1 --kernel-include-regex "k_integrate" -d $ROOT/$O/att -o att -- \
  python $ROOT/bench.py --res 512 --color 0 --steps 2 --warmup 1 --extras 0 --cpu-baseline 0 --host-path 0 > $ROOT/$O/att_stdout.txt 2> $ROOT/$O/att_stderr.txt
echo "att rc=$?" | tee $ROOT/$O/att_rc.txt
ls -laR $ROOT/$O/att 2>/dev/null | head -40 >> $ROOT/$O/att_rc.txt
# (b)
timeout 120 rocprofv3 -L > $ROOT/$O/counters_avail.txt 2>&1
grep -o -E "\b(SQ|SQC|TA|TCP|TCC|TD|GRBM|SPI)_[A-Za-z0-9_]+" $ROOT/$O/counters_avail.txt | sort -u > $ROOT/$O/counter_names.txt
wc -l $ROOT/$O/counter_names.txt
have() { for c in "$@"; do grep -q -x "$c" $ROOT/$O/counter_names.txt && echo -n "$c "; done; }
BENCH="python $ROOT/bench.py --warmup 2 --cpu-baseline 0 --scene-b 0 --host-path 0 --extras 0 --steps 5"
pass_() {  # name, bench args, counters...
  local name=$1 args=$2; shift 2
  local cs=$(have "$@")
  [ -z "$cs" ] && { echo "$name: no counter of this set exists here"; return; }
  timeout 300 rocprofv3 --pmc $cs -d $ROOT/$O/pmc_$name -o pmc --output-format csv -- $BENCH $args > $ROOT/$O/bench_$name.json 2> $ROOT/$O/bench_$name.err
  echo "$name rc=$? [$cs]"
}
for KEY in c0 c1; do
  [ $KEY = c0 ] && A="--color 0" || A="--color 1"
  if [ $KEY = c1 ]; then  # the headline has its SQ1 / SQ2 passes from round 5 (same kernel source): only the new sets
    pass() { case $1 in *_SQ1|*_SQ2|*_SQ5|*_TCP2|*_TCC) return;; esac; pass_ "$@"; }
  else
    pass() { pass_ "$@"; }
  fi
  pass ${KEY}_SQ1 "$A" SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE
  pass ${KEY}_SQ2 "$A" SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD TCC_HIT_sum TCC_MISS_sum
  pass ${KEY}_SQ3 "$A" SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH
  pass ${KEY}_SQ4 "$A" SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_IFETCH SQ_INSTS_VALU_MFMA_I8 SQ_THREAD_CYCLES_VALU SQ_INSTS_SENDMSG
  pass ${KEY}_SQ5 "$A" SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_IFETCH_LEVEL SQ_ACCUM_PREV_HIRES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
  pass ${KEY}_SQC "$A" SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_MISSES_DUPLICATE
  pass ${KEY}_TA "$A" TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum
  pass ${KEY}_TCP "$A" TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum
  pass ${KEY}_TCP2 "$A" TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum
  pass ${KEY}_TCC "$A" TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_TAG_STALL_sum TCC_BUSY_sum TCC_EA0_WRREQ_STALL_sum
done
cd $ROOT
for d in $O/pmc_*; do python tools/pmc_reduce.py $d > $O/summary_$(basename $d).json 2>> $O/reduce.err; done
find $O -name "*.csv" -size +1M -delete
du -sh $O; ls $O | head -80
