"""MI355X (gfx950) MSM / NTT backend for the halo2 delay-encryption prover.

Host-side mirror (Python) of the upstream interface the reference reaches through
``create_proof`` (benches/delay_enc.rs:123-131): ``arithmetic.best_multiexp`` /
``arithmetic.best_fft``, ``domain.EvaluationDomain`` and ``commitment.Params``
(ParamsKZG / ParamsIPA ``commit`` / ``commit_lagrange``), all thin wrappers over the
C ABI in ``include/dehalo.h`` (``libdehalo.so``, hand-written HIP).  There is no CPU
path: importing works anywhere, but creating a :class:`Context` without the built
library or without a gfx950 device raises.

The directory name has a hyphen, so load it with ``__graft_entry__.load_package()``
(module name ``dehalo2_amd``).
"""
from . import fields  # noqa: F401
from ._lib import Context, DehaloError, Bases, library_path, load_library  # noqa: F401
from .arithmetic import batch_invert, best_fft, best_multiexp, eval_polynomial, grand_product  # noqa: F401
from .domain import EvaluationDomain  # noqa: F401
from .commitment import Params, ParamsIPA  # noqa: F401
from . import evaluation, lookup  # noqa: F401

__all__ = ["Context", "DehaloError", "Bases", "best_multiexp", "best_fft", "eval_polynomial", "batch_invert", "grand_product", "EvaluationDomain", "Params", "ParamsIPA", "fields",
           "library_path", "load_library"]
