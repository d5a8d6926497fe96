#!/usr/bin/env python3
"""Developer tool (no GPU needed): instruction mix of the hot basic blocks of one kernel, from hipcc's gfx950 assembly.
    tools/isa_mix.py attention.hip attention_lds_kernelILi2ELi2ELb0EE [min MFMAs per block]"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, key = sys.argv[1], sys.argv[2]
min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 16
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-S", "--cuda-device-only",
                    os.path.join(ROOT, "av-separation-transformer_amd", "csrc", src), "-o", out], check=True, stderr=subprocess.DEVNULL)
    s = open(out).read()
i = s.index(key); i = s.index(":", i); j = s.index(".Lfunc_end", i)
blocks, cur, name = [], [], "entry"
for l in s[i:j].split("\n"):
    if re.match(r"^\.LBB\d+_\d+:", l):
        blocks.append((name, cur)); name, cur = l.split(":")[0], []
    else:
        cur.append(l.strip())
blocks.append((name, cur))
print(f"# {src} :: {key}: basic blocks with >= {min_mfma} MFMAs (one pass through each = one 16-key tile of the loop)")
for name, ins in blocks:
    ins = [x for x in ins if x and not x.startswith((";", "."))]
    if sum("v_mfma" in x for x in ins) < min_mfma:
        continue
    c = collections.Counter()
    for x in ins:
        op = x.split()[0]
        c["mfma" if op.startswith("v_mfma") else "v_exp" if op.startswith("v_exp") else "v_pk" if op.startswith("v_pk_") else
          "permlane" if op.startswith("v_permlane") else "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else
          "waitcnt" if op.startswith("s_waitcnt") else "barrier" if op.startswith("s_barrier") else "salu" if op.startswith("s_") else
          "vmem" if op.startswith(("global_", "buffer_")) else op] += 1
    print(f"{name}: {len(ins)} instructions: " + ", ".join(f"{k} {v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1])))
