"""Lists, per basic block of one gemm_bf16_8ph_kernel instantiation, the full memory drains (s_waitcnt vmcnt(0)) and
scratch accesses -- a spilled value reloaded inside the tile loop costs such a drain.
usage: python tools/gemm_asm_drains.py file.s [template-args, default 'ILb1ELb1ELb0ELi0ELb0E']"""
import re
import sys

path = sys.argv[1]
inst = sys.argv[2] if len(sys.argv) > 2 else "ILb1ELb1ELb0ELi0ELb0E"
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if "gemm_bf16_8ph_kernel" + inst in l and l.startswith("_ZN"))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
meta = [l.strip() for l in lines if False]
blk, info, order = "entry", {}, []
for i in range(start, end):
    l = lines[i]
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blk = m.group(1)
    if blk not in info:
        info[blk] = dict(first=i + 1, vm0=[], scr=[], st=0, mfma=0, dma=0, loop="Loop" in l)
        order.append(blk)
    t = l.strip()
    d = info[blk]
    if t == "s_waitcnt vmcnt(0)":
        d["vm0"].append(i + 1)
    elif t.startswith("scratch_"):
        d["scr"].append(f"{i + 1}:{t.split()[0]}")
    elif t.startswith("global_store"):
        d["st"] += 1
    elif "v_mfma" in t:
        d["mfma"] += 1
    elif "global_load_lds" in t:
        d["dma"] += 1
print(f"function lines {start + 1}..{end}")
for b in order:
    d = info[b]
    if d["scr"] or (d["vm0"] and (d["mfma"] or d["dma"] or d["st"] >= 4)):
        print(b, "line", d["first"], "mfma", d["mfma"], "dma", d["dma"], "stores", d["st"], "vmcnt(0) at", d["vm0"], "scratch", d["scr"])
for l in lines[end:]:
    if "gemm_bf16_8ph_kernel" + inst in l and ".name:" in l:
        j = lines.index(l)
        print([x.strip() for x in lines[j:j + 12] if "spill" in x or "vgpr_count" in x or "private_segment" in x])
        break
