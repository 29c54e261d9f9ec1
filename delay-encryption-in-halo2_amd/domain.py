"""halo2_proofs::poly::EvaluationDomain mirror (halo2_proofs/src/poly/domain.rs @ v2023_04_20;
SURVEY.md A.3): lagrange_to_coeff, coeff_to_extended, extended_to_coeff."""
from __future__ import annotations

import numpy as np

from ._lib import Context
from .fields import FieldSpec


class EvaluationDomain:
    def __init__(self, ctx: Context, field: FieldSpec, j: int, k: int):
        """``j`` = maximum gate degree, ``k`` = log2 rows (EvaluationDomain::new(j, k))."""
        self.ctx, self.field, self.k, self.j = ctx, field, k, j
        p = field.p
        self.n = 1 << k
        self.quotient_poly_degree = j - 1
        ek = k
        while (1 << ek) < self.n * self.quotient_poly_degree:
            ek += 1
        self.extended_k = ek
        if ek > field.two_adicity:
            raise ValueError("extended_k exceeds the field's two-adicity")
        root = field.root_of_unity
        ext_omega = root
        for _ in range(field.two_adicity - ek):
            ext_omega = ext_omega * ext_omega % p
        omega = ext_omega
        for _ in range(ek - k):
            omega = omega * omega % p
        self.omega, self.omega_inv = omega, pow(omega, -1, p)
        self.extended_omega, self.extended_omega_inv = ext_omega, pow(ext_omega, -1, p)
        self.ifft_divisor = pow(self.n % p, -1, p)
        self.extended_ifft_divisor = pow((1 << ek) % p, -1, p)
        self.g_coset = field.zeta
        self.g_coset_inv = self.g_coset * self.g_coset % p
        e = field.encode
        self._omega_inv_m, self._ifft_div_m = e(self.omega_inv), e(self.ifft_divisor)
        self._ext_omega_m, self._ext_omega_inv_m = e(self.extended_omega), e(self.extended_omega_inv)
        self._ext_div_m, self._zeta_m = e(self.extended_ifft_divisor), e(self.g_coset)

    def extended_len(self) -> int:
        return 1 << self.extended_k

    def lagrange_to_coeff(self, a) -> np.ndarray:
        return self.ctx.intt_scaled(self.field.id, a, self.k, self._omega_inv_m, self._ifft_div_m)

    def coeff_to_extended(self, a) -> np.ndarray:
        return self.ctx.coset_ntt(self.field.id, a, self.k, self.extended_k, self._ext_omega_m, self._zeta_m)

    def extended_to_coeff(self, a) -> np.ndarray:
        out = self.ctx.coset_intt(self.field.id, a, self.extended_k, self._ext_omega_inv_m, self._ext_div_m, self._zeta_m)
        return out[: self.n * self.quotient_poly_degree]
