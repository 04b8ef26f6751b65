#!/usr/bin/env python3
"""Static instruction mix of the kernels in a gfx950 assembly file (hipcc -save-temps / -S output).

usage: scripts/isa_mix.py file.s [name-substring ...]

Per kernel: VGPR / AGPR / SGPR / scratch / LDS from its .amdhsa_kernel block, and the static count of instructions
per class in its body (MFMA, other vector ALU, LDS, vector memory, scalar ALU, scalar memory, waits, barriers,
branches).  Static counts: a rolled loop's body counts once -- read them next to the kernel's loop structure.
The ratio "valu_per_mfma" is the figure profiles/*/instruction_mix.json reports dynamically from the SQ counters.
"""
import re
import sys


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma"
    if op.startswith(("ds_", "lds_")):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("s_load", "s_buffer_load", "s_store", "s_memtime", "s_memrealtime", "s_dcache")):
        return "smem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_call")):
        return "branch"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_"):
        return "salu"
    return "other"


def kernels(text):
    meta = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
        blk = m.group(2)

        def g(k, d="?"):
            q = re.search(k + r" (\S+)", blk)
            return q.group(1) if q else d
        meta[m.group(1)] = {"vgpr": g(r"\.amdhsa_next_free_vgpr"), "accum_offset": g(r"\.amdhsa_accum_offset"),
                            "sgpr": g(r"\.amdhsa_next_free_sgpr"), "scratch": g(r"\.amdhsa_private_segment_fixed_size"),
                            "lds": g(r"\.amdhsa_group_segment_fixed_size")}
    for name, md in meta.items():
        i = text.find("\n" + name + ":")
        if i < 0:
            continue
        body = text[i:]
        j = body.find(".Lfunc_end")
        body = body[:j if j > 0 else None]
        cnt, ops = {}, {}
        for ln in body.splitlines():
            ln = ln.split(";")[0].strip()
            if not ln or ln.endswith(":") or ln.startswith("."):
                continue
            op = ln.split()[0]
            c = classify(op)
            cnt[c] = cnt.get(c, 0) + 1
            if c == "valu":
                ops[op] = ops.get(op, 0) + 1
        yield name, md, cnt, ops


def demangle(names):
    import subprocess
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), text=True, stdout=subprocess.PIPE).stdout.splitlines()
        return dict(zip(names, out))
    except OSError:
        return {n: n for n in names}


def main():
    text = open(sys.argv[1]).read()
    pats = sys.argv[2:]
    rows = list(kernels(text))
    dm = demangle([r[0] for r in rows])
    for name, md, cnt, ops in rows:
        pretty = re.sub(r"\(.*\)$", "", dm[name]).replace("void ", "").replace("asep::", "")
        if pats and not any(p in pretty for p in pats):
            continue
        mf = max(cnt.get("mfma", 0), 1)
        print(f"{pretty}\n   vgpr {md['vgpr']} (accum at {md['accum_offset']}) sgpr {md['sgpr']} scratch {md['scratch']} lds {md['lds']}")
        print("   " + "  ".join(f"{k} {cnt.get(k, 0)}" for k in ("mfma", "valu", "lds", "vmem", "salu", "smem", "wait", "barrier", "branch", "nop"))
              + f"   valu_per_mfma {cnt.get('valu', 0) / mf:.2f}")
        top = sorted(ops.items(), key=lambda kv: -kv[1])[:14]
        print("   valu: " + ", ".join(f"{k} {v}" for k, v in top))


if __name__ == "__main__":
    main()
