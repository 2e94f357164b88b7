cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
rm -rf /tmp/pm
timeout 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pm -- python $GRAFT_REPO_ROOT/tools/probe_refine_group.py 98304 250 200 200 > /tmp/pm.log 2>/tmp/pm.err
f=$(find /tmp/pm -name '*counter_collection.csv' | head -1)
python - "$f" <<PY
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"]
    if "refine_group_gemm" not in k and "refine_exact_wide" not in k: continue
    k=k.split("(")[0][-40:]
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
    n[(k,r["Counter_Name"])]+=1
for k,v in acc.items():
    print(k, {c: "%.3g"%(x/n[(k,c)]) for c,x in v.items()})
PY
done
