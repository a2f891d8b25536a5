#!/bin/bash
export TRK_LAB=1   # tools are lab runs: lab knobs are honoured (trtools_amd/_knobs.py)
# SQ instruction-mix counters of every kernel of a command (one rocprofv3 --pmc pass per counter group; no other
# trace domains).  usage: tools/sq_counters.sh <out.txt> <command...>
out=$1; shift
cd /tmp && export TMPDIR=/tmp
: > $out
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
         "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  rm -rf /tmp/pm_sq
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/pm_sq -o x --output-format csv -- "$@" > /dev/null 2>&1
  python3 - >> $out <<PY
import csv, glob, collections, re
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/pm_sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+(<[^>]*>)?)", r["Kernel_Name"])
        k = (m.group(1) if m else r["Kernel_Name"])[:60].replace(" ", "")
        acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print("%-60s %-24s avg %.4g  n %d" % (k, c, sum(v) / len(v), len(v)))
PY
done
