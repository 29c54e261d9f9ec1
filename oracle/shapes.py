"""TEST INFRASTRUCTURE (only tests/, __graft_entry__.smoke() and bench.py's checker leg may import this).

The constraint systems of the reference's circuits as plain data, written down by the checker on its own -- the same tuple the product's
`plonk.ConstraintSystem.description()` produces, so that the CPU restatement's keygen / create_proof / verify_proof do not take the shape they check
from the code they check.  Sources: the reference's configure() (src/lib.rs:126-160: one MainGate::configure, one RangeChip::configure with
composition_bit_lens [8, 1, 8, 4] / overflow_bit_lens [0, 0, 6], restated in SURVEY.md Appendix C) and, for the column and query ORDER inside
MainGate / RangeChip, [UPSTREAM halo2wrong maingate v2023_04_20] from memory (the crates are not in the container: parity unpinned there, see
DESIGN.md section 5).

    description = (num_advice, num_fixed, num_instance, gates, lookups, permutation_columns, advice_queries, fixed_queries, instance_queries, minimum_degree)
    expression  = ("advice" | "fixed" | "instance", column, rotation) | ("sum", a, b) | ("product", a, b)
"""

# MainGate's fixed columns, in the order configure() creates them: sa, sb, sc, sd, se, s_mul_ab, s_mul_cd, s_next_e, s_constant
SA, SB, SC, SD, SE, S_MUL_AB, S_MUL_CD, S_NEXT_E, S_CONSTANT = range(9)
# RangeChip's: the table (tag, value), the two tag columns, the two selectors (complex selectors become fixed columns)
T_TAG, T_VALUE, TAG_COMPOSITION, TAG_OVERFLOW, S_COMPOSITION, S_OVERFLOW = range(9, 15)


def _adv(c, r=0):
    return ("advice", c, r)


def _fix(c):
    return ("fixed", c, 0)


def _mul(a, b):
    return ("product", a, b)


def _sum(terms):
    acc = terms[0]
    for t in terms[1:]:
        acc = ("sum", acc, t)
    return acc


def maingate_description(range_lookups: bool = True) -> tuple:
    """MainGate (+ RangeChip): 5 advice a..e, 9 (+ 6) fixed, 1 instance; the one gate
        a sa + b sb + c sc + d sd + e se + a b s_mul_ab + c d s_mul_cd + e_next s_next_e + s_constant = 0;
    five lookups (composition on a, b, c, d; overflow on a): (tag, selector * advice) in (t_tag, t_value); equality on a..e and the instance column."""
    a, b, c, d, e = (_adv(i) for i in range(5))
    gate = _sum([_mul(a, _fix(SA)), _mul(b, _fix(SB)), _mul(c, _fix(SC)), _mul(d, _fix(SD)), _mul(e, _fix(SE)),
                 _mul(_mul(a, b), _fix(S_MUL_AB)), _mul(_mul(c, d), _fix(S_MUL_CD)), _mul(_adv(4, 1), _fix(S_NEXT_E)), _fix(S_CONSTANT)])
    lookups = ()
    fixed_queries = [(i, 0) for i in range(9)]
    if range_lookups:
        table = (_fix(T_TAG), _fix(T_VALUE))
        lookups = tuple(((_fix(TAG_COMPOSITION), _mul(_fix(S_COMPOSITION), _adv(col))), table) for col in range(4))
        lookups += (((_fix(TAG_OVERFLOW), _mul(_fix(S_OVERFLOW), _adv(0))), table),)
        # queries in the order the lookups' closures make them: selector, tag, (advice), table tag, table value -- first use only
        fixed_queries += [(S_COMPOSITION, 0), (TAG_COMPOSITION, 0), (T_TAG, 0), (T_VALUE, 0), (S_OVERFLOW, 0), (TAG_OVERFLOW, 0)]
    permutation_columns = tuple(("advice", i) for i in range(5)) + (("instance", 0),)
    advice_queries = tuple((i, 0) for i in range(5)) + ((4, 1),)
    return (5, 15 if range_lookups else 9, 1, (gate,), lookups, permutation_columns, advice_queries, tuple(fixed_queries), ((0, 0),), 0)
