#!/usr/bin/env python3
"""Rows per gadget of the MainGate / RangeChip layouter (witness.py; csrc/witness.hip writes the same rows) and the totals of the reference's circuits beside the
row counts the reference publishes (benches/README.md:56-99): the table behind DESIGN.md section 5.  CPU only.   python tools/witness_rows.py"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package()
from dehalo2_amd import witness as W
p = pkg.fields.BN254_FR.p
rnd = random.Random(1)
n_big = rnd.getrandbits(2048) | (1 << 2047) | 1
x = rnd.getrandbits(2040)


def rows_of(fn):
    lay = W.Layouter(p)
    chip = W.BigIntChip(lay)
    state = fn(lay, chip, True, None)
    r0 = lay.rows
    fn(lay, chip, False, state)
    return lay.rows - r0


def gadget(setup_fn, run_fn):
    return lambda lay, chip, setup, state: setup_fn(lay, chip) if setup else run_fn(lay, chip, state)


three = lambda lay, chip: (chip.assign_integer(x), chip.assign_integer(x >> 3), chip.assign_integer(n_big))
gadgets = [
    ("RangeChip::assign of one 64-bit limb (eight 8-bit limbs)", gadget(lambda l, c: None, lambda l, c, s: c.range_limb(x & (2**64 - 1)))),
    ("RangeChip::assign of one 70-bit carry (eight 8-bit limbs + a 6-bit overflow limb)", gadget(lambda l, c: None, lambda l, c, s: l.range_assign(x & (2**70 - 1), 8, 70))),
    ("is_equal / is_zero / select / assert_equal", gadget(lambda l, c: (l.assign_value(5), l.assign_value(5), l.assign_bit(1)),
                                                         lambda l, c, s: (l.is_equal(s[0], s[1]), l.is_zero(s[0]), l.select(s[0], s[1], s[2]), l.assert_equal(s[0], s[1])))),
    ("div_mod_main_gate", gadget(lambda l, c: (l.assign_value(x & (2**100 - 1)), l.assign_constant(1 << 64)), lambda l, c, s: c.div_mod_main_gate(*s))),
    ("assign_integer (32 range-checked limbs)", gadget(lambda l, c: None, lambda l, c, s: c.assign_integer(x))),
    ("mul 32 x 32 limbs (63 product limbs)", gadget(three, lambda l, c, s: c.mul(s[0], s[1]))),
    ("add 32 + 32 limbs", gadget(three, lambda l, c, s: c.add(s[0], s[1]))),
    ("sub (a + max - b, selects, two sub_unchecked)", gadget(three, lambda l, c, s: c.sub(s[2], s[1]))),
    ("assert_in_field (x < n)", gadget(three, lambda l, c, s: c.assert_in_field(s[1], s[2]))),
    ("mul_mod (quotient + remainder limbs, two mul, carried equality)", gadget(three, lambda l, c, s: c.mul_mod(s[0], s[1], s[2]))),
    ("to_bits of a 5-bit exponent limb", gadget(lambda l, c: l.assign_value(21), lambda l, c, s: l.to_bits(s, 5))),
    ("one Poseidon permutation (T = 5, R_F = 8, R_P = 57)", gadget(lambda l, c: W.PoseidonChip(l, W.poseidon_spec(p), [l.assign_value(i) for i in range(5)]), lambda l, c, s: s.permutation([]))),
]
print("rows per gadget (32 limbs of 64 bits, 2048-bit modulus):")
for name, fn in gadgets:
    print("  %-84s %6d" % (name, rows_of(fn)))
print("closed forms: mul_mod %d, RSA region 1,860 + 7,981 bits + ceil(bits / 4) = %s, hash region %d, cipher region (2 words, keyed by the digest) %d" %
      (W.mul_mod_rows(), [W.rsa_region_rows(b) for b in (1, 2, 15)], W.hash_region_rows(), W.cipher_region_rows(2, True)))
print("totals of whole circuits | the reference's README (the published figure is the last used row's index: rows - 1):")
published = {2: 17822, 3: 25803, 5: 41766, 8: 65709, 15: 121578}
for bits in (1, 2, 3, 5):
    e = (1 << (bits - 1)) | 1
    _, info = W.mod_pow_witness(p, 18, n_big, e, x, bits)
    print("  mod_pow, %2d-bit exponent: %7d rows | %s" % (bits, info.total_rows, published.get(bits, "-")))
for bits in (8, 15):
    print("  mod_pow, %2d-bit exponent: %7d rows (closed form) | %s" % (bits, W.rsa_region_rows(bits), published[bits]))
for bits, pub in ((2, 26461), (15, 130248)):
    total = W.rsa_region_rows(bits) + W.hash_region_rows() + W.cipher_region_rows(2, True)
    print("  delay_enc, %2d-bit exponent: %7d rows (closed form) | %d: +5,035 = the hash region of the revision the table was made with (DESIGN.md section 5)%s" %
          (bits, total, pub, "; the published run drew e = 0 (31 rows fewer)" if bits == 2 else ""))
for msg in (1, 2, 3, 4, 5, 16, 31):
    _, info = W.pose_enc_witness(p, 13, [5, 6], [0] * msg)
    print("  pose_enc, %2d message word(s): %6d rows | %d" % (msg, info.total_rows, {1: 1446, 2: 1450, 3: 1454, 4: 1458, 5: 2180, 16: 3660, 31: 6592}[msg]))
