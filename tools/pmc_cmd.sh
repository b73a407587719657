#!/bin/bash
# PMC counters of ONE kernel inside an arbitrary command: separate rocprofv3 --pmc passes (never combined with other
# trace domains), averages per dispatch record written to the output file.
#   tools/pmc_cmd.sh <kernel name substring> <out.txt> <command ...>
set -u
SUB=$1; OUT=$2; shift 2
export TMPDIR=/tmp
db() { ls "$1"/*/*_results.db 2>/dev/null | head -1; }
echo "# kernel *$SUB* in: $*" > "$OUT"
echo "# rocprofv3 --pmc <group> --kernel-trace, one pass per group; averages per dispatch record (SQ_*: per XCD/SE record;" >> "$OUT"
echo "# FETCH_SIZE / WRITE_SIZE in KB per dispatch, FETCH_SIZE x2 on gfx950 for 16-byte streaming reads)" >> "$OUT"
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf /tmp/pc_$i
  rocprofv3 --pmc $grp --kernel-trace -d /tmp/pc_$i -- "$@" > /tmp/pc_$i.log 2>&1
  python tools/pmc_kernel.py "$SUB" "$(db /tmp/pc_$i)" >> "$OUT"
done
python tools/prof_summary.py "$(db /tmp/pc_1)" | grep "$SUB" >> "$OUT"
cat "$OUT"
