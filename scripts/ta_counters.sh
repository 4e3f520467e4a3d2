cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "TA_[A-Z_a-z0-9]*\|TCP_[A-Z_a-z0-9]*\|TD_[A-Z_a-z0-9]*" | sort -u | tr '\n' ' ' > /root/repo/gpurun_out/counters_list.txt
for grp in "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_ACCESSES_sum" "GRBM_GUI_ACTIVE TA_BUSY_avr TA_BUSY_max"; do
  n=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /root/repo/gpurun_out/ta/pass_$n -- python /root/repo/bench.py --no-cpu-baseline --steps 16 --warmup 40 > /root/repo/gpurun_out/ta/$n.log 2>&1
done
python /root/repo/scripts/pmc_summary.py /root/repo/gpurun_out/ta > /root/repo/gpurun_out/ta/summary.json
