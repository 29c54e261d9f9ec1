// hostrng.hpp -- sources of the prover's random scalars (blinding rows, blinds, the vanishing argument's random polynomial), the
// `rng: R` argument of create_proof [UPSTREAM halo2_proofs/src/plonk/prover.rs; the reference passes OsRng, benches/delay_enc.rs:128].
//   DEHALO_RNG_OS        the default: 32 bytes of operating-system entropy (getrandom) key a ChaCha20 stream per call; every scalar is
//                        256 stream bits masked to the modulus' bit length and rejected when >= p -- uniform over the field (the blinding
//                        rows on the host; the random polynomial's n scalars by a ChaCha20 KERNEL under the same key, prover.hip);
//   DEHALO_RNG_PCG64     TESTS / BENCHMARKS ONLY (not a CSPRNG): numpy's PCG64 stream from a given state, four 64-bit outputs per scalar,
//                        top word masked to 61 bits -- the stream dehalo2_amd.prover.SeededRng and the CPU restatement consume, so that
//                        proofs can be compared byte for byte;
//   DEHALO_RNG_CALLBACK  the caller's own generator.
// A scalar is handed over as a Montgomery REPRESENTATION (multiplication by R is a bijection of the field: uniform stays uniform).
#pragma once
#include <sys/random.h>

#include <cstdint>
#include <cstring>

#include "../../include/dehalo.h"
#include "hostfield.hpp"

typedef unsigned __int128 u128;

struct Pcg64 {
    u128 state, inc;
    static u128 mult() { return ((u128)0x2360ED051FC65DA4ULL << 64) | 0x4385DF649FCCF645ULL; }
    uint64_t next() {
        state = state * mult() + inc;
        const uint64_t hi = (uint64_t)(state >> 64), lo = (uint64_t)state;
        const uint64_t x = hi ^ lo;
        const unsigned rot = (unsigned)(hi >> 58);
        return (x >> rot) | (x << ((64 - rot) & 63));
    }
    void advance(u128 delta) {
        u128 acc_mult = 1, acc_plus = 0, cur_mult = mult(), cur_plus = inc;
        while (delta) {
            if (delta & 1) {
                acc_mult *= cur_mult;
                acc_plus = acc_plus * cur_mult + cur_plus;
            }
            cur_plus = (cur_mult + 1) * cur_plus;
            cur_mult *= cur_mult;
            delta >>= 1;
        }
        state = acc_mult * state + acc_plus;
    }
};

struct ChaCha20 {
    uint32_t st[16];
    static uint32_t rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
    void init(const uint8_t key[32], uint64_t stream) {
        static const uint32_t sigma[4] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
        memcpy(st, sigma, 16);
        memcpy(st + 4, key, 32);
        st[12] = st[13] = 0;                        // 64-bit block counter
        st[14] = (uint32_t)stream;
        st[15] = (uint32_t)(stream >> 32);
    }
    void block(uint32_t out[16]) {
        uint32_t x[16];
        memcpy(x, st, 64);
#define CC_QR(a, b, c, d)                                                                                   \
    x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16); x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 12);                 \
    x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8);  x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 7);
        for (int i = 0; i < 10; i++) {
            CC_QR(0, 4, 8, 12) CC_QR(1, 5, 9, 13) CC_QR(2, 6, 10, 14) CC_QR(3, 7, 11, 15)
            CC_QR(0, 5, 10, 15) CC_QR(1, 6, 11, 12) CC_QR(2, 7, 8, 13) CC_QR(3, 4, 9, 14)
        }
#undef CC_QR
        for (int i = 0; i < 16; i++) out[i] = x[i] + st[i];
        if (++st[12] == 0) ++st[13];
    }
};

// One proof's generator.  `position` counts scalars in upstream's draw order; fork(skip) gives an independent generator positioned
// `skip` scalars further on (the helper thread's: it draws the random polynomial while the earlier phases run).
struct HostRng {
    int kind = DEHALO_RNG_OS;
    const HostField* f = nullptr;
    Pcg64 pcg{};
    ChaCha20 cc{};
    uint32_t ccbuf[16];
    int ccpos = 16;
    uint8_t key[32];
    dehalo_rng_fill_fn fill_fn = nullptr;
    void* user = nullptr;
    uint64_t position = 0;
    dehalo_rng* caller = nullptr;      // PCG64: the caller's state is advanced when the proof is done

    int init(const dehalo_rng* r, const HostField* field) {
        f = field;
        kind = r ? r->kind : DEHALO_RNG_OS;
        position = 0;
        if (kind == DEHALO_RNG_OS) {
            size_t got = 0;
            while (got < 32) {
                const ssize_t n = getrandom(key + got, 32 - got, 0);
                if (n <= 0) return DEHALO_ERR_UNSUPPORTED;
                got += (size_t)n;
            }
            cc.init(key, 0);
            ccpos = 16;
            return 0;
        }
        if (kind == DEHALO_RNG_PCG64) {
            pcg.state = ((u128)r->pcg_state[1] << 64) | r->pcg_state[0];
            pcg.inc = ((u128)r->pcg_inc[1] << 64) | r->pcg_inc[0];
            return 0;
        }
        if (kind == DEHALO_RNG_CALLBACK && r->fill) {
            fill_fn = r->fill;
            user = r->user;
            return 0;
        }
        return DEHALO_ERR_INVALID;
    }
    HostRng fork(uint64_t skip, uint64_t stream) const {
        HostRng o = *this;
        o.position = position + skip;
        if (kind == DEHALO_RNG_PCG64) o.pcg.advance((u128)4 * skip);
        if (kind == DEHALO_RNG_OS) {      // an independent ChaCha20 stream under the same key
            o.cc.init(key, stream);
            o.ccpos = 16;
        }
        return o;
    }
    void skip(uint64_t count) {
        position += count;
        if (kind == DEHALO_RNG_PCG64) pcg.advance((u128)4 * count);
    }
    uint64_t cc64() {
        if (ccpos >= 16) {
            cc.block(ccbuf);
            ccpos = 0;
        }
        const uint64_t v = (uint64_t)ccbuf[ccpos] | ((uint64_t)ccbuf[ccpos + 1] << 32);
        ccpos += 2;
        return v;
    }
    int scalars(uint64_t* out, size_t count) {
        if (kind == DEHALO_RNG_PCG64) {
            for (size_t i = 0; i < count; i++) {
                for (int j = 0; j < 4; j++) out[4 * i + j] = pcg.next();
                out[4 * i + 3] &= ((uint64_t)1 << 61) - 1;
            }
        } else if (kind == DEHALO_RNG_OS) {
            const uint64_t mask = f->bits >= 256 ? ~(uint64_t)0 : (((uint64_t)1 << (f->bits - 192)) - 1);
            for (size_t i = 0; i < count; i++) {
                uint64_t* o = out + 4 * i;
                do {
                    for (int j = 0; j < 4; j++) o[j] = cc64();
                    o[3] &= mask;
                } while (HostField::geq(o, f->p));
            }
        } else {
            const int rc = fill_fn(user, out, count, position);
            if (rc != 0) return DEHALO_ERR_INVALID;
        }
        position += count;
        return 0;
    }
};
