// blake2b.hpp -- BLAKE2b (RFC 7693) with a personalisation string: the hash behind halo2's Blake2bWrite / Challenge255
// [UPSTREAM halo2_proofs/src/transcript.rs: Blake2bParams::new().hash_length(64).personal(b"Halo2-Transcript")] and behind
// vk.transcript_repr ("Halo2-Verify-Key").  Unkeyed, sequential mode.  Host side only.
#pragma once
#include <cstdint>
#include <cstring>

struct Blake2b {
    uint64_t h[8];
    uint64_t t[2];
    uint8_t buf[128];
    size_t buflen;
    size_t outlen;

    static uint64_t rotr(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
    static uint64_t load64(const uint8_t* p) {
        uint64_t v;
        memcpy(&v, p, 8);      // little-endian host (x86-64)
        return v;
    }

    void init(size_t digest_len, const char personal[16]) {
        static const uint64_t IV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                                       0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
        uint8_t param[64];
        memset(param, 0, sizeof param);
        param[0] = (uint8_t)digest_len;   // digest length
        param[1] = 0;                     // key length
        param[2] = 1;                     // fanout
        param[3] = 1;                     // depth
        if (personal) memcpy(param + 48, personal, 16);
        for (int i = 0; i < 8; i++) h[i] = IV[i] ^ load64(param + 8 * i);
        t[0] = t[1] = 0;
        buflen = 0;
        outlen = digest_len;
        memset(buf, 0, sizeof buf);
    }

    void compress(const uint8_t block[128], bool last) {
        static const uint64_t IV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                                       0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
        static const uint8_t SIGMA[12][16] = {
            {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
            {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
            {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
            {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
            {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
            {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
        uint64_t m[16], v[16];
        for (int i = 0; i < 16; i++) m[i] = load64(block + 8 * i);
        for (int i = 0; i < 8; i++) v[i] = h[i];
        for (int i = 0; i < 8; i++) v[8 + i] = IV[i];
        v[12] ^= t[0];
        v[13] ^= t[1];
        if (last) v[14] = ~v[14];
#define B2B_G(a, b, c, d, x, y)       \
    v[a] = v[a] + v[b] + (x);         \
    v[d] = rotr(v[d] ^ v[a], 32);     \
    v[c] = v[c] + v[d];               \
    v[b] = rotr(v[b] ^ v[c], 24);     \
    v[a] = v[a] + v[b] + (y);         \
    v[d] = rotr(v[d] ^ v[a], 16);     \
    v[c] = v[c] + v[d];               \
    v[b] = rotr(v[b] ^ v[c], 63);
        for (int r = 0; r < 12; r++) {
            const uint8_t* s = SIGMA[r];
            B2B_G(0, 4, 8, 12, m[s[0]], m[s[1]])
            B2B_G(1, 5, 9, 13, m[s[2]], m[s[3]])
            B2B_G(2, 6, 10, 14, m[s[4]], m[s[5]])
            B2B_G(3, 7, 11, 15, m[s[6]], m[s[7]])
            B2B_G(0, 5, 10, 15, m[s[8]], m[s[9]])
            B2B_G(1, 6, 11, 12, m[s[10]], m[s[11]])
            B2B_G(2, 7, 8, 13, m[s[12]], m[s[13]])
            B2B_G(3, 4, 9, 14, m[s[14]], m[s[15]])
        }
#undef B2B_G
        for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[8 + i];
    }

    void update(const void* data, size_t len) {
        const uint8_t* in = (const uint8_t*)data;
        while (len) {
            if (buflen == 128) {          // the buffer is only compressed once MORE input arrives: the last block must stay for final()
                t[0] += 128;
                if (t[0] < 128) t[1]++;
                compress(buf, false);
                buflen = 0;
            }
            size_t take = 128 - buflen;
            if (take > len) take = len;
            memcpy(buf + buflen, in, take);
            buflen += take;
            in += take;
            len -= take;
        }
    }

    // digest of everything absorbed so far; the state itself is left untouched (works on a copy), so the transcript can go on
    void digest(uint8_t* out) const {
        Blake2b c = *this;
        c.t[0] += c.buflen;
        if (c.t[0] < c.buflen) c.t[1]++;
        memset(c.buf + c.buflen, 0, 128 - c.buflen);
        c.compress(c.buf, true);
        uint8_t full[64];
        memcpy(full, c.h, 64);
        memcpy(out, full, outlen);
    }
};
