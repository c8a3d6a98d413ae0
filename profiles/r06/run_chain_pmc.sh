# Counters of the receive chain's tail kernels (one-stream chain of bench.py --chain-only; under --pmc every kernel runs alone).
#   gpurun --timeout 1500 -- 'sh profiles/r06/run_chain_pmc.sh <tag>'
# Two separate --pmc passes (never combined with other trace domains); summary -> gpurun_out/<tag>/chain_tail_counters.md
TAG=${1:-r06pmc}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $O/pmc1 $O/pmc2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --chain-only"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $O/pmc1 -o p -- $B > $O/pmc1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc2 -o p -- $B > $O/pmc2.log 2>&1
python - $O <<'PY' > $O/chain_tail_counters.md
import csv, glob, sys, collections
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for d in ("pmc1", "pmc2"):
    f = glob.glob(f"{O}/{d}/**/*counter_collection.csv", recursive=True)
    if not f:
        print("no counter file in", d); continue
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "anonymous" not in k or "k_fused" in k: continue
        name = k.split("(anonymous namespace)::")[-1].split("(")[0]
        key = (name, r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", ""))
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[key] = (r.get("VGPR_Count", r.get("Arch_VGPR_Count", "?")), r.get("LDS_Block_Size", "?"))
print("| kernel (grid) | launches | waves | VGPRs | LDS B | cycles (GRBM/8) | us @ 2.4 GHz | VALU instr / wave | SALU / wave | LDS / wave | VMEM rd / wr per wave | VALU pipes busy | waiting (share of wave cycles) | LDS conflict share |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
for key in sorted(acc, key=lambda k: (k[0], int(k[1] or 0))):
    c = {n: sum(v[-8:]) / len(v[-8:]) for n, v in acc[key].items()}
    w = c.get("SQ_WAVES", 0) or 1
    cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8.0
    busy = c.get("SQ_ACTIVE_INST_VALU", 0) * 4 / 1024 / cyc if cyc else 0
    wait = c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else 0
    conf = c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"] if c.get("SQ_LDS_IDX_ACTIVE") else 0
    print(f"| `{key[0]}` ({key[1]}) | {len(next(iter(acc[key].values())))} | {w:.0f} | {meta[key][0]} | {meta[key][1]} | {cyc:.3g} | {cyc/2400:.1f} | {c.get('SQ_INSTS_VALU',0)/w:.0f} | {c.get('SQ_INSTS_SALU',0)/w:.0f} | {c.get('SQ_INSTS_LDS',0)/w:.0f} | {c.get('SQ_INSTS_VMEM_RD',0)/w:.0f} / {c.get('SQ_INSTS_VMEM_WR',0)/w:.0f} | {busy:.2f} | {wait:.2f} | {conf:.2f} |")
PY
rm -rf $O/pmc1 $O/pmc2
cat $O/chain_tail_counters.md
