#!/bin/bash
# PMC counters of ONE kernel of the relation-query workload (BASELINE C2), separate rocprofv3 --pmc passes (never
# combined with other trace domains); averages per dispatch record written to the output file.
#   tools/pmc_rq_kernel.sh <kernel name substring> <out.txt> [extra bench.py arguments]
set -u
SUB=$1; OUT=$2; shift 2
export TMPDIR=/tmp
db() { ls "$1"/*/*_results.db 2>/dev/null | head -1; }
echo "# kernel *$SUB* in: python bench.py --workload rq --steps 3 --warmup 1 --no-cpu-baseline --no-parity $*" > "$OUT"
echo "# rocprofv3 --pmc <group> --kernel-trace, one pass per group; averages per dispatch record (SQ_*: per XCD/SE record;" >> "$OUT"
echo "# FETCH_SIZE / WRITE_SIZE in KB per dispatch, FETCH_SIZE x2 on gfx950 for 16-byte streaming reads)" >> "$OUT"
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf /tmp/pk_$i
  rocprofv3 --pmc $grp --kernel-trace -d /tmp/pk_$i -- python bench.py --workload rq --steps 3 --warmup 1 --no-cpu-baseline --no-parity "$@" > /tmp/pk_$i.log 2>&1
  python tools/pmc_kernel.py "$SUB" "$(db /tmp/pk_$i)" >> "$OUT"
done
python tools/prof_summary.py "$(db /tmp/pk_1)" | grep "$SUB" >> "$OUT"
cat "$OUT"
