"""Instruction mix of one kernel in a hipcc -S listing: python isa_mix.py file.s <substring>"""
import re, sys, collections
lines = open(sys.argv[1]).read().splitlines()
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(key) + r"\S*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
ops = collections.Counter()
for l in lines[start:end]:
    m = re.match(r"\s+([a-z][a-z_0-9]+)\s", l)
    if m:
        ops[m.group(1)] += 1
print(lines[start].split(":")[0], "static instructions:", sum(ops.values()))
groups = collections.Counter()
for k, v in ops.items():
    g = ("div_seq" if k in ("v_div_scale_f64", "v_div_fmas_f64", "v_div_fixup_f64", "v_rcp_f64_e32", "v_rcp_f64") else
         "f64_arith" if re.match(r"v_(fma|mul|add|max|min)_f64", k) else
         "scratch" if k.startswith("scratch_") else "lds" if k.startswith("ds_") else
         "global" if k.startswith("global_") else "cmp/cndmask" if re.match(r"v_(cmp|cndmask)", k) else
         "salu" if k.startswith("s_") else "other_valu")
    groups[g] += v
for k, v in groups.most_common():
    print(f"  {k:14s} {v}")
for k, v in ops.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 30):
    print(f"    {k:26s} {v}")
