"""halo2_proofs::plonk::lookup::prover mirrors (halo2_proofs/src/plonk/lookup/prover.rs @ v2023_04_20;
SURVEY.md 8(f) row 2): the permuted (input, table) pair of one lookup argument."""
from __future__ import annotations

import numpy as np

from ._lib import Context, DehaloError
from .fields import FieldSpec


class ConstraintSystemFailure(Exception):
    """upstream's Error::ConstraintSystemFailure: an input value is not in the table."""


def permute_expression_pair(ctx: Context, field: FieldSpec, input_expression, table_expression, usable_rows: int):
    """-> (permuted_input, permuted_table), usable_rows x 4 u64 each (Montgomery); the caller appends
    the blinding rows.  Raises ConstraintSystemFailure like upstream returns Err(..)."""
    try:
        return ctx.permute_expression_pair(field.id, input_expression, table_expression, usable_rows)
    except DehaloError as e:
        if e.code == -6:
            raise ConstraintSystemFailure(str(e)) from None
        raise
