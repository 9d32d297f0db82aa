#!/usr/bin/env python3
"""Listing-level report of one gfx950 kernel from `hipcc --save-temps` assembly: the method behind the round-3 wins of the AHC round kernel
 as a tool.  It answers the three questions that paid:

  1. where does the memory counter get drained?   every `s_waitcnt vmcnt(N)` in program order with the vector-memory operations issued since
     the last full drain (static, straight-line view: a wait behind STORES, or behind a request a rare branch left pending, waits for them);
  2. what is issued in front of the first request? (scalar loads of by-value kernel arguments and their `s_waitcnt lgkmcnt(0)`);
  3. how many instructions of which unit sit between the barriers?

usage:  hipcc -O3 --offload-arch=gfx950 --save-temps -c file.hip        (writes file-hip-amdgcn-amd-amdhsa-gfx950.s)
        isa_report.py file-hip-amdgcn-amd-amdhsa-gfx950.s 'ahc_round_tILb0' [--listing out.lst] [--from IDX] [--to IDX]
"""
import argparse
import re
from collections import Counter


def kernel_body(text, pattern):
    names = re.findall(r"^(\S+):\s*(?:;.*)?$", text, re.M)
    cands = [n for n in names if re.search(pattern, n) and not n.startswith(".")]
    if not cands:
        raise SystemExit(f"no kernel matching {pattern!r}")
    name = cands[0]
    i = text.index("\n" + name + ":")
    j = text.index(".end_amdhsa_kernel", i)
    return name, text[i:j]


def strip_preload_header(body):
    """With -amdgpu-kernarg-preload-count the kernel starts with a compatibility header (s_load of the preloaded arguments, a wait, s_branch,
    .p2align 8) for firmware without preloading; hardware that preloads enters 256 bytes further on and never runs it."""
    m = re.search(r"\n\s*s_branch\s+\S+\s*\n\s*\.p2align\s+8\s*\n", body)
    if m and m.start() < 2000 and "s_load" in body[:m.start()] and "v_" not in body[:m.start()].split(":", 1)[-1]:
        head = body[:body.index("\n") + 1]
        return head + body[m.end():], True
    return body, False


def listing(body):
    out, k = [], 0
    for line in body.split("\n"):
        t = line.split(";")[0].strip()
        if not t:
            continue
        if re.match(r"^\.?[A-Za-z_0-9$]+:$", t):
            out.append((None, t))
            continue
        if t.startswith("."):
            continue
        out.append((k, t))
        k += 1
    return out


def unit(op):
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("ds_",)):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith(("s_load", "s_buffer_load", "s_memtime", "s_memrealtime")):
        return "smem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_nop"):
        return "nop"
    return "salu"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("kernel")
    ap.add_argument("--listing")
    ap.add_argument("--from", dest="lo", type=int, default=0)
    ap.add_argument("--to", dest="hi", type=int, default=10 ** 9)
    a = ap.parse_args()
    text = open(a.asm).read()
    name, body = kernel_body(text, a.kernel)
    body, preload = strip_preload_header(body)
    lst = listing(body)
    if a.listing:
        with open(a.listing, "w") as f:
            for k, t in lst:
                f.write(t + "\n" if k is None else f"{k:6d}  {t}\n")
    meta = {m.group(1): m.group(2) for m in re.finditer(r"; (NumVgprs|NumAgprs|ScratchSize|Occupancy|NumSgprs): (\d+)", text[text.index(name + ":"):text.index(name + ":") + len(body) + 4000])}
    print(f"kernel {name[:100]}\n  {sum(1 for k, _ in lst if k is not None)} instructions; {meta}")
    if preload:
        print("  (kernel-argument preload header skipped: the listing starts at the entry point the hardware uses)")
    # 1. vmcnt drains
    print("\nvector-memory counter (static program order; L = loads, S = stores / atomics issued since the last vmcnt(0)):")
    loads = stores = 0
    first_vmem = None
    for k, t in lst:
        if k is None or not (a.lo <= k <= a.hi):
            continue
        op = t.split()[0]
        u = unit(op)
        if u == "vmem":
            if first_vmem is None:
                first_vmem = k
            if "load" in op and "atomic" not in op:
                loads += 1
            else:
                stores += 1
        m = re.match(r"s_waitcnt .*vmcnt\((\d+)\)", t)
        if m:
            n = int(m.group(1))
            flag = "   <-- drains everything, STORES included" if n == 0 and stores else ""
            print(f"  {k:6d}  vmcnt({n:2d})   since last drain: {loads} L, {stores} S{flag}")
            if n == 0:
                loads = stores = 0
    # 2. prologue
    print(f"\nin front of the first vector-memory request (instruction {first_vmem}):")
    pro = Counter()
    for k, t in lst:
        if k is None or first_vmem is None or k >= first_vmem:
            continue
        op = t.split()[0]
        pro[unit(op)] += 1
        if unit(op) == "smem" or (op == "s_waitcnt" and "lgkmcnt" in t):
            print(f"  {k:6d}  {t}")
    print("  ", dict(pro))
    # 3. per-barrier sections
    print("\ninstructions between workgroup barriers (static):")
    sec, start = Counter(), 0
    for k, t in lst:
        if k is None:
            continue
        op = t.split()[0]
        if unit(op) == "barrier":
            print(f"  {start:6d} .. {k:6d}  {dict(sec)}")
            sec, start = Counter(), k + 1
        else:
            sec[unit(op)] += 1
    print(f"  {start:6d} .. end     {dict(sec)}")


if __name__ == "__main__":
    main()
