#!/bin/bash
# PMC counters summed per kernel-name substring: pmc_kernel.sh OUTDIR SUBSTR[,SUBSTR...] -- program args...
# (six counter sets, each in its own rocprofv3 --kernel-trace --pmc pass)
out=$1; pat=$2; shift 3
case $out in /*) ;; *) out=$PWD/$out ;; esac
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -o c -- "$@" > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/p$i.log; }
  python3 - "$out/p$i/c_counter_collection.csv" "$pat" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.Counter()
pats = sys.argv[2].split(",")
try:
    for r in csv.DictReader(open(sys.argv[1])):
        for p in pats:
            if p in r["Kernel_Name"]:
                acc[p, r["Counter_Name"]] += float(r["Counter_Value"]); n[p, r["Counter_Name"]] += 1
except Exception as e:
    print("no csv:", e)
for (p, k) in sorted(acc): print(f"{p:24s} {k:32s} {acc[p, k]:16.0f} total over {n[p, k]} launches")
PY
  rm -rf $out/p$i
done
