// san_rx_bank.cpp -- the C++ face of the receive chain (sdrpp-tetra-demodulator_amd/host/tetra_rx_bank.h) under AddressSanitizer +
// UBSan (TEST TOOL).  Without arguments: the calls that need no GPU work (bad configurations, a bank that was never initialised)
// return statuses.  With `gpu <C> <N> <calls> <iq.bin> <out.bin>`: streams <calls> blocks of C x N complex64 samples (channel major,
// one after the other in the file) through TetraRxBank::process, fetches every kind of every call the way a consumer would (the
// previous call's blocks while the next one runs), and dumps them for the Python test to compare with RxChain's.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "tetra_rx_bank.h"

using dsp::demod::TetraRxBank;
using dsp::demod::TetraRxMultiBank;

static int fail(const char* what, int rc) { std::fprintf(stderr, "san_rx_bank: %s -> %d\n", what, rc); return 1; }

int main(int argc, char** argv) {
    if (argc < 2 || std::strcmp(argv[1], "gpu") != 0) {
        tetra_rx_config_t cfg;
        if (tetra_rx_default_config(&cfg) != TETRA_OK) return fail("default_config", -1);
        TetraRxBank bank;
        cfg.kinds = 1 << 9;                                             // no such kind
        if (bank.init(cfg) != TETRA_ERR_ARG) return fail("init with a bad kind mask", 0);
        cfg.kinds = 0; cfg.flags = 64;
        if (bank.init(cfg) != TETRA_ERR_ARG) return fail("init with a bad flag", 0);
        TetraRxBank::Blocks b;
        std::vector<tetra_lmac_cell_state_t> cells;
        if (bank.fetch(TETRA_RX_KIND_SB1, b) != TETRA_ERR_ARG || bank.fetch(17, b) != TETRA_ERR_ARG) return fail("fetch without a handle", 0);
        if (bank.wait() != TETRA_ERR_ARG || bank.reset() != TETRA_ERR_ARG || bank.process(4, nullptr) != TETRA_ERR_ARG) return fail("calls without a handle", 0);
        if (tetra_rx_type1_bits(TETRA_RX_KIND_SCH_F) != 268 || tetra_rx_type1_bits(-1) != TETRA_ERR_ARG) return fail("type1_bits", 0);
        TetraRxMultiBank multi;
        tetra_rx_default_config(&cfg);
        cfg.demod.n_channels = 1;
        if (multi.init(cfg, std::vector<int>{ 0, 0 }) != TETRA_ERR_ARG || multi.init(cfg, std::vector<int>()) != TETRA_ERR_ARG) return fail("multi init", 0);
        if (multi.shards() != 0 || multi.wait() != TETRA_OK) return fail("empty multi bank", 0);
        std::printf("san_rx_bank: ok\n");
        return 0;
    }
    if (argc != 7 && argc != 8) return fail("usage", -1);
    const int C = std::atoi(argv[2]), N = std::atoi(argv[3]), calls = std::atoi(argv[4]);
    std::vector<float> iq((size_t)2 * C * N);
    std::FILE* in = std::fopen(argv[5], "rb");
    std::FILE* out = std::fopen(argv[6], "wb");
    if (!in || !out) return fail("files", -1);
    tetra_rx_config_t cfg;
    tetra_rx_default_config(&cfg);
    cfg.demod.n_channels = C;
    cfg.demod.max_samples = N;
    const int n_shards = argc == 8 ? std::atoi(argv[7]) : 0;             // > 0: TetraRxMultiBank with that many shards, all on device 0
    TetraRxBank bank;
    TetraRxMultiBank multi;
    int rc = n_shards ? multi.init(cfg, std::vector<int>((size_t)n_shards, 0)) : bank.init(cfg);
    if (rc != TETRA_OK) return fail("init", rc);
    auto dump = [&](int which) -> int {
        for (int k = 0; k < TETRA_RX_N_KINDS; k++) {
            TetraRxBank::Blocks b;
            const int r = n_shards ? multi.fetch(k, b, which) : bank.fetch(k, b, which);
            if (r != TETRA_OK) return r;
            const int32_t hdr[3] = { k, (int32_t)b.info.size(), b.bitsPerBlock };
            std::fwrite(hdr, sizeof(hdr), 1, out);
            std::fwrite(b.info.data(), sizeof(tetra_rx_block_t), b.info.size(), out);
            std::fwrite(b.type1.data(), 1, b.type1.size(), out);
        }
        return TETRA_OK;
    };
    for (int k = 0; k < calls; k++) {
        if (std::fread(iq.data(), sizeof(float), iq.size(), in) != iq.size()) return fail("short input", k);
        rc = n_shards ? multi.process(N, iq.data()) : bank.process(N, iq.data());
        if (rc != TETRA_OK) return fail("process", rc);
        if (k >= 1 && (rc = dump(1)) != TETRA_OK) return fail("fetch previous", rc);      // call k - 1's blocks while call k runs
    }
    if ((rc = dump(0)) != TETRA_OK) return fail("fetch last", rc);
    if ((rc = n_shards ? multi.wait() : bank.wait()) != TETRA_OK) return fail("wait", rc);
    std::vector<tetra_lmac_cell_state_t> cells;
    std::vector<tetra_bsync_state_t> sync;
    if (n_shards) {
        if ((rc = multi.cells(cells)) != TETRA_OK) return fail("cells", rc);
        int first = -1, count = -1;
        multi.shardInfo(n_shards - 1, first, count);
        if (first + count != C || multi.channels() != C || multi.shards() != n_shards) return fail("shardInfo", first + count);
    } else if ((rc = bank.cells(cells)) != TETRA_OK || (rc = bank.syncStates(sync)) != TETRA_OK) return fail("cells", rc);
    std::fwrite(cells.data(), sizeof(tetra_lmac_cell_state_t), cells.size(), out);
    if (n_shards ? multi.reset() != TETRA_OK : (bank.setParam(TETRA_PARAM_AGC_RATE, 0.02) != TETRA_OK || bank.reset() != TETRA_OK)) return fail("setParam / reset", -1);
    std::fclose(in);
    std::fclose(out);
    std::printf("san_rx_bank: ok\n");
    return 0;
}
