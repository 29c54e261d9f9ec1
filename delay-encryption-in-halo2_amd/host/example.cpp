// example.cpp -- the reference's bench flow (benches/pose_enc.rs:41-135: ParamsKZG::setup when no params file is cached, else read it; keygen_vk /
// keygen_pk, write and re-read the proving key, create_proof into a Blake2bWrite transcript) from C++ over the C ABI, with no interpreter anywhere.
//
//   example                 build / link check: one tiny best_fft (needs a GPU to run)
//   example <dir>           <dir>/params.bin   ParamsKZG RawBytes; when it does not exist it is MADE (ParamsKZG::setup on the device) from
//                                              <dir>/secret.bin (32 B: the scalar the reference would draw from OsRng, Montgomery form) and written
//                           <dir>/circuit.bin  u32 k | fixed columns (9 x 2^k x 32 B, Montgomery) | permutation mapping (6 x 2^k u64) |
//                                              advice (5 x 2^k x 32 B, Montgomery) | transcript_repr (32 B) | PCG64 state, inc (2 x 16 B)
//                           -> <dir>/pk.bin (ProvingKey RawBytes), <dir>/vk.bin, <dir>/proof.bin
// The circuit shape is the MainGate of PoseidonEncCircuit (5 advice, 9 fixed, 1 instance; halo2wrong maingate [UPSTREAM]) built here
// with the ConstraintSystem builder; tests/test_native.py writes the inputs, runs this program and checks the proof it wrote.
#include <cstdio>
#include <cstring>
#include <fstream>
#include <memory>

#include "halo2_backend.hpp"

using namespace halo2_amd;

static std::vector<uint8_t> slurp(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + path);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static void dump(const std::string& path, const std::vector<uint8_t>& b) {
    std::ofstream f(path, std::ios::binary);
    f.write((const char*)b.data(), (std::streamsize)b.size());
    if (!f) throw std::runtime_error("cannot write " + path);
}

// MainGate::configure: a sa + b sb + c sc + d sd + e se + a b s_mul_ab + c d s_mul_cd + e(next) s_next + s_constant = 0
static void maingate(ConstraintSystem& cs) {
    for (uint32_t i = 0; i < 5; i++) cs.enable_equality(DEHALO_COLUMN_ADVICE, i);
    cs.enable_equality(DEHALO_COLUMN_INSTANCE, 0);
    ConstraintSystem::Expr adv[5], fx[9];
    for (uint32_t i = 0; i < 5; i++) adv[i] = cs.query_advice(i);
    const ConstraintSystem::Expr e_next = cs.query_advice(4, 1);
    for (uint32_t i = 0; i < 9; i++) fx[i] = cs.query_fixed(i);
    ConstraintSystem::Expr g = cs.product(adv[0], fx[0]);
    for (uint32_t i = 1; i < 5; i++) g = cs.sum(g, cs.product(adv[i], fx[i]));
    g = cs.sum(g, cs.product(cs.product(adv[0], adv[1]), fx[5]));
    g = cs.sum(g, cs.product(cs.product(adv[2], adv[3]), fx[6]));
    g = cs.sum(g, cs.product(e_next, fx[7]));
    g = cs.sum(g, fx[8]);
    cs.create_gate({g});
}

int main(int argc, char** argv) {
    try {
        Backend be(0);
        if (argc < 2) {
            std::vector<Fe> a(4, Fe{0, 0, 0, 0});
            Fe one_m{0x34786d38fffffffdULL, 0x992c350be41914adULL, 0xffffffffffffffffULL, 0x3fffffffffffffffULL};  // pasta::Fp R
            be.best_fft(DEHALO_FIELD_PASTA_FP, a, one_m, 2);
            std::printf("%s ok\n", dehalo_version());
            return 0;
        }
        const std::string dir = argv[1];
        const std::vector<uint8_t> c = slurp(dir + "/circuit.bin");
        uint32_t k;
        memcpy(&k, c.data(), 4);
        std::unique_ptr<ParamsKZG> params_owner;
        if (std::ifstream(dir + "/params.bin", std::ios::binary)) params_owner.reset(new ParamsKZG(be, DEHALO_CURVE_BN254_G1, slurp(dir + "/params.bin")));
        else {      // benches/pose_enc.rs:44-54: no cached file -> setup, write
            const std::vector<uint8_t> sb = slurp(dir + "/secret.bin");
            if (sb.size() != 32) throw std::runtime_error("secret.bin: 32 bytes expected");
            Fe s;
            memcpy(s.data(), sb.data(), 32);
            params_owner.reset(new ParamsKZG(be, DEHALO_CURVE_BN254_G1, k, s));
            dump(dir + "/params.bin", params_owner->write());
        }
        ParamsKZG& params = *params_owner;
        const size_t n = (size_t)1 << k;
        if (c.size() != 4 + 9 * n * 32 + 6 * n * 8 + 5 * n * 32 + 32 + 32) throw std::runtime_error("circuit.bin: unexpected size");
        std::vector<Fe> fixed(9 * n), advice(5 * n);
        std::vector<uint64_t> mapping(6 * n);
        const uint8_t* p = c.data() + 4;
        memcpy(fixed.data(), p, 9 * n * 32); p += 9 * n * 32;
        memcpy(mapping.data(), p, 6 * n * 8); p += 6 * n * 8;
        memcpy(advice.data(), p, 5 * n * 32); p += 5 * n * 32;
        Fe repr;
        memcpy(repr.data(), p, 32); p += 32;
        dehalo_rng rng{};
        rng.kind = DEHALO_RNG_PCG64;
        memcpy(rng.pcg_state, p, 16);
        memcpy(rng.pcg_inc, p + 16, 16);

        ConstraintSystem cs(5, 9, 1);
        maingate(cs);
        {
            ProvingKey fresh(be, params, cs, fixed, mapping);      // keygen_vk + keygen_pk
            dump(dir + "/pk.bin", fresh.write());
            dump(dir + "/vk.bin", fresh.vk_write());
        }
        ProvingKey pk(be, DEHALO_CURVE_BN254_G1, cs, slurp(dir + "/pk.bin"), 0);      // ... cached on disk, read back (benches/pose_enc.rs:103-115)
        pk.set_transcript_repr(repr);
        Backend side(0);
        Prover prover(be, params, pk, &side);
        Blake2bWrite transcript(DEHALO_CURVE_BN254_G1);
        prover.create_proof(advice, {{}}, &rng, transcript);
        const std::vector<uint8_t> proof = transcript.finalize();
        dump(dir + "/proof.bin", proof);
        std::printf("k = %u: proof of %zu bytes written\n", k, proof.size());
    } catch (const std::exception& e) {
        std::printf("error: %s\n", e.what());
        return 1;
    }
    return 0;
}
