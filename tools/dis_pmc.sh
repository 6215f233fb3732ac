#!/bin/bash
# PMC passes over the DPSelect distance kernel at the BASELINE size (tools/dis_check.py launches it 13 times).
out=gpurun_out/${1:-dis_pmc}
mkdir -p $out
export TMPDIR=/tmp
for pass in "fetch FETCH_SIZE" "sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM"; do
  set -- $pass; name=$1; shift
  timeout 300 rocprofv3 --pmc $@ -d $out/$name -o $name --output-format csv -- python3 tools/dis_check.py > $out/$name.log 2>&1 < /dev/null
done
python tools/pmc_summary.py $out/pmc_dis.csv $(find $out/fetch $out/sq1 $out/sq2 -name "*counter_collection.csv") 2>&1 | tail -1
rm -rf $out/fetch $out/sq1 $out/sq2
python - <<PY
import csv
for r in csv.DictReader(open("$out/pmc_dis.csv")):
    if "dis_kernel" in r["kernel"]:
        for k, v in r.items(): print(k, v)
PY
