"""halo2_proofs::plonk::evaluation mirrors (halo2_proofs/src/plonk/evaluation.rs @ v2023_04_20;
SURVEY.md 8(f) row 1): the GraphEvaluator container, built the way upstream builds it
(add_constant / add_rotation / add_calculation with de-duplication), compiled once into a device
program and run over device-resident extended-domain columns; plus the permutation and lookup
argument terms of Evaluator::evaluate_h.  Field values are canonical Python ints at this level
and 4 x u64 Montgomery limbs at the C ABI."""
from __future__ import annotations

from dataclasses import dataclass, field as dc_field
import numpy as np
from typing import List, Optional, Sequence, Tuple

from ._lib import Context
from .fields import FieldSpec

# ValueSource kinds / Calculation ops (dehalo_source_kind, dehalo_calc_op)
CONSTANT, INTERMEDIATE, FIXED, ADVICE, INSTANCE, CHALLENGE, BETA, GAMMA, THETA, Y, PREVIOUS = range(11)
ADD, SUB, MUL, SQUARE, DOUBLE, NEGATE, HORNER, STORE = range(8)

Source = Tuple[int, int, int]          # (kind, index, rotation index)

# dehalo.h: optional device-internal element form
FORM_OUT_INTERNAL, FORM_IN_INTERNAL = 1, 2
COLUMNS_INTERNAL, VALUES_INTERNAL = 1, 2


@dataclass
class GraphEvaluator:
    """constants start as [0, 1, 2] like upstream's GraphEvaluator::default()."""
    constants: List[int] = dc_field(default_factory=lambda: [0, 1, 2])
    rotations: List[int] = dc_field(default_factory=list)
    calculations: List[tuple] = dc_field(default_factory=list)      # (op, a, b, parts, target)
    num_intermediates: int = 0

    def add_rotation(self, rotation: int) -> int:
        if rotation in self.rotations:
            return self.rotations.index(rotation)
        self.rotations.append(rotation)
        return len(self.rotations) - 1

    def add_constant(self, constant: int) -> Source:
        if constant in self.constants:
            return (CONSTANT, self.constants.index(constant), 0)
        self.constants.append(constant)
        return (CONSTANT, len(self.constants) - 1, 0)

    def add_calculation(self, op: int, a: Source, b: Source = (CONSTANT, 0, 0), parts: Sequence[Source] = ()) -> Source:
        key = (op, a, b, tuple(parts))
        for c in self.calculations:
            if (c[0], c[1], c[2], tuple(c[3])) == key:
                return (INTERMEDIATE, c[4], 0)
        target = self.num_intermediates
        self.calculations.append((op, a, b, tuple(parts), target))
        self.num_intermediates += 1
        return (INTERMEDIATE, target, 0)

    def column(self, kind: int, index: int, rotation: int = 0) -> Source:
        return (kind, index, self.add_rotation(rotation))

    # ---- device side ----
    def compile(self, ctx: Context, field: FieldSpec) -> "CompiledGraph":
        parts: List[Source] = []
        calcs = []
        for op, a, b, pp, target in self.calculations:
            calcs.append((op, a, b, len(parts), len(pp), target))
            parts.extend(pp)
        consts = field.encode_many(self.constants) if self.constants else []
        handle = ctx.graph_create(field.id, consts, self.rotations, calcs, parts, self.num_intermediates)
        return CompiledGraph(ctx, field, handle)


class CompiledGraph:
    def __init__(self, ctx: Context, field: FieldSpec, handle):
        self.ctx, self.field, self.handle = ctx, field, handle

    def evaluate_device(self, fixed: Sequence[int], advice: Sequence[int], instance: Sequence[int], challenges: Sequence[int], beta: Optional[int],
                        gamma: Optional[int], theta: Optional[int], y: Optional[int], log_rows: int, rot_scale: int, d_previous: int, d_out: int,
                        stream: int = 0, form_flags: int = 0, ctx: Optional[Context] = None):
        """Column arguments are device pointers (extended-domain cosets, 1 << log_rows elements).  `ctx`: run on another context
        of the same device (compiled graphs may be shared between contexts, dehalo.h)."""
        e = self.field.encode
        enc = lambda v: v if v is None or isinstance(v, np.ndarray) else e(v)           # (already encoded values pass through)
        if isinstance(challenges, np.ndarray):
            ch = challenges if challenges.shape[0] else None
        else:
            ch = self.field.encode_many(list(challenges)) if challenges else None
        (ctx or self.ctx).graph_evaluate_device(self.handle, fixed, advice, instance, ch, enc(beta), enc(gamma), enc(theta), enc(y), log_rows, rot_scale,
                                       d_previous, d_out, stream, form_flags)

    def release(self):
        if self.handle is not None:
            self.ctx.graph_release(self.handle)
            self.handle = None


def permutation_h_device(ctx: Context, field: FieldSpec, z: Sequence[int], columns: Sequence[int], sigma: Sequence[int], chunk_len: int, last_rotation: int,
                         l0: int, l_last: int, l_active_row: int, beta: int, gamma: int, y: int, delta: int, zeta: int, extended_omega: int, log_rows: int,
                         rot_scale: int, d_values: int, stream: int = 0, form_flags: int = 0):
    """Evaluator::evaluate_h's permutation terms folded into d_values in place (device pointers)."""
    if len(columns) != len(sigma):
        raise ValueError("permutation: columns.len() != cosets.len()")
    e = field.encode
    ctx.permutation_h_device(field.id, list(z), list(columns), list(sigma), chunk_len, last_rotation, l0, l_last, l_active_row, e(beta), e(gamma), e(y), e(delta),
                             e(beta * zeta % field.p), e(extended_omega), log_rows, rot_scale, d_values, stream, form_flags)


def lookup_h_device(ctx: Context, field: FieldSpec, product: int, permuted_input: int, permuted_table: int, table_value: int, l0: int, l_last: int,
                    l_active_row: int, beta: int, gamma: int, y: int, log_rows: int, rot_scale: int, d_values: int, stream: int = 0, form_flags: int = 0):
    """Evaluator::evaluate_h's five terms of one lookup argument folded into d_values in place."""
    e = field.encode
    ctx.lookup_h_device(field.id, product, permuted_input, permuted_table, table_value, l0, l_last, l_active_row, e(beta), e(gamma), e(y), log_rows, rot_scale,
                        d_values, stream, form_flags)


def lookup_h_batch_device(ctx: Context, field: FieldSpec, lookups, l0: int, l_last: int, l_active_row: int, beta: int, gamma: int, y: int, log_rows: int,
                          rot_scale: int, d_values: int, stream: int = 0, form_flags: int = 0):
    """Evaluator::evaluate_h's lookup loop in one pass: `lookups` = (product, permuted_input, permuted_table, table_value) device
    pointers per lookup argument, in order; each lookup's five terms are folded into d_values as lookup_h_device folds them."""
    e = field.encode
    ctx.lookup_h_batch_device(field.id, list(lookups), l0, l_last, l_active_row, e(beta), e(gamma), e(y), log_rows, rot_scale, d_values, stream, form_flags)
