import sys, os, io
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as entry
pkg = entry.load_package(); po, co = entry.load_oracle()
import plonk_oracle as PO, pairing as pr
from dehalo2_amd import circuits, keygen, native
k, rl = 6, False
curve = po.BN254
circ = circuits.synthesize(curve.scalar.p, k, rl, seed=3)
srs = PO.setup_srs(curve, k, 0x1234, 4)
ctx = pkg.Context(0)
params = keygen.ParamsKZG(ctx, pkg.fields.BN254, k, srs["g"], srs["g_lagrange"])
pk = keygen.keygen(ctx, params, circ.cs, circ.fixed, circ.assembly, circ.selectors)
b = io.BytesIO(); pk.write(b); py = b.getvalue()
np_ = native.ParamsKZG.create(ctx, pkg.fields.BN254, k, srs["g"], srs["g_lagrange"])
npk = native.ProvingKey.keygen(ctx, np_, circ.cs, circ.fixed, circ.assembly, circ.selectors)
nat = npk.write()
print(len(py), len(nat))
n, m = 1 << k, 2 << k
cs = circ.cs
nf, npc = cs.num_fixed, 6
off = 8 + 64 * (nf + npc)
secs = [("vk", off)]
for nm in ("l0", "l_last", "l_active"): secs.append((nm, 4 + 32 * m))
for nm, cnt, ln in (("fixed_values", nf, n), ("fixed_polys", nf, n), ("fixed_cosets", nf, m), ("perm_values", npc, n), ("perm_polys", npc, n), ("perm_cosets", npc, m)):
    secs.append((nm, 4 + cnt * (4 + 32 * ln)))
o = 0
for nm, ln in secs:
    a, c = py[o:o + ln], nat[o:o + ln]
    nd = sum(1 for i in range(0, ln) if a[i] != c[i])
    print(nm, ln, "same" if a == c else "DIFF %d bytes, first at %d" % (nd, next(i for i in range(ln) if a[i] != c[i])))
    o += ln
# expected l0 coset from the oracle
F = PO.Fld(curve.scalar)
key = PO.keygen(curve, srs, cs.description(), k, circ.fixed, circ.assembly.mapping, 4)
print(key.keys())
def sec(buf, name):
    o = 0
    for nm, ln in secs:
        if nm == name: return buf[o:o + ln]
        o += ln
want = np.ascontiguousarray(key["l0"]).tobytes() if hasattr(key["l0"], "tobytes") else None
print(type(key["l0"]), getattr(key["l0"], "shape", None))
a, c = sec(py, "l0")[4:], sec(nat, "l0")[4:]
print("python == oracle:", a == want, " native == oracle:", c == want)
# is the native one a permutation of the python one?
ea = [a[i:i+32] for i in range(0, len(a), 32)]; ec = [c[i:i+32] for i in range(0, len(c), 32)]
print("same multiset:", sorted(ea) == sorted(ec))
idx = [ea.index(x) if x in ea else -1 for x in ec[:12]]
print("native[i] = python[j]:", idx)
p = curve.scalar.p
ia = [int.from_bytes(x, "little") for x in ea]; ic = [int.from_bytes(x, "little") for x in ec]
print("canonical?", all(v < p for v in ia), all(v < p for v in ic), max(ic) >> 250)
for i in range(6):
    print(i, hex(ia[i])[:20], hex(ic[i])[:20], hex(ic[i] * pow(ia[i], -1, p) % p)[:24], (ic[i] - ia[i]) % p == 0)
