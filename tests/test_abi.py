"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol include/*.h declares."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "tetra_demod.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tetra_demod_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_all_exported(pkg):
    L = pkg.load_library()
    names = _declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(L, n), "declared in include/tetra_demod.h but not exported: " + n
    assert set(pkg.binding.EXPORTS) == set(names)
    assert L.tetra_demod_abi_version() == 6
    assert L.tetra_demod_build_id().decode() == pkg.build.source_hash() == pkg.build.lib_build_id()


def test_channeliser_header_symbols_all_exported(pkg):
    src = open(os.path.join(ROOT, "include", "tetra_chan.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = sorted(set(re.findall(r"\b(tetra_chan_[a-z0-9_]+)\s*\(", src)))
    L = pkg.load_library()
    assert len(names) == 11 and set(names) == set(pkg.chan_binding.CHAN_EXPORTS)
    for n in names:
        assert hasattr(L, n), n


def test_burst_scan_header_symbols_all_exported(pkg):
    src = open(os.path.join(ROOT, "include", "tetra_burst_scan.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = sorted(set(re.findall(r"\b(tetra_find_train_seq_batch[a-z_]*|tetra_ts_indicator_[a-z_]+)\s*\(", src)))
    L = pkg.load_library()
    assert set(names) == set(pkg.scan_binding.SCAN_EXPORTS) | {"tetra_ts_indicator_destroy"}
    for n in names:
        assert hasattr(L, n), n


def test_lmac_header_symbols_all_exported_and_host_helpers(pkg):
    """include/tetra_lmac.h: every declared entry point is exported; the two host-only helpers (no GPU needed) give the
    reference's block parameters (tetra_lower_mac.c:58-105) and scrambling code (tetra_scramb.c:87-99)."""
    src = open(os.path.join(ROOT, "include", "tetra_lmac.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = sorted(set(re.findall(r"\b(tetra_lmac_[a-z0-9_]+)\s*\(", src)))
    L = pkg.load_library()
    assert set(names) == set(pkg.lmac_binding.LMAC_EXPORTS)
    for n in names:
        assert hasattr(L, n), n
    want = {0: (120, 80, 60, 11, 1), 1: (216, 144, 124, 101, 1), 2: (216, 144, 124, 101, 1), 3: (30, 30, 14, 0, 0),
            4: (168, 112, 92, 13, 1), 5: (432, 288, 268, 103, 1)}
    for t, w in want.items():
        p = pkg.lmac_binding.blk_param(t)
        assert (p.type345_bits, p.type2_bits, p.type1_bits, p.interleave_a, p.have_crc16) == w
    from tests.emul import lmac_emul_bind
    for t, w in lmac_emul_bind.BLK_PARAM.items():
        assert w == want[t][:4]
    assert pkg.lmac_binding.scramb_init(262, 1, 5) == 0x41800117      # value of the reference's tetra_scramb_get_init
    from oracle import ref_binding
    if ref_binding.lmac_available():
        for mcc, mnc, cc in ((262, 1, 5), (1023, 16383, 63), (0, 0, 0), (234, 14, 1)):
            assert pkg.lmac_binding.scramb_init(mcc, mnc, cc) == ref_binding.scramb_get_init(mcc, mnc, cc)


def test_burst_sync_header_symbols_all_exported(pkg):
    src = open(os.path.join(ROOT, "include", "tetra_burst_sync.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = sorted(set(re.findall(r"\b(tetra_bsync_[a-z0-9_]+|tetra_burst_demux_[a-z_]*device|tetra_burst_index_device)\s*\(", src)))
    L = pkg.load_library()
    assert set(names) == set(pkg.bsync_binding.BSYNC_EXPORTS)
    for n in names:
        assert hasattr(L, n), n
    assert C.sizeof(pkg.bsync_binding.BsyncState) == 16


def test_default_config_is_the_plugins(pkg):
    cfg = pkg.binding.default_config()
    # src/main.cpp:35-44,78-84
    assert (cfg.symbolrate, cfg.samplerate, cfg.rrc_tap_count) == (18000, 36000, 65)
    assert abs(cfg.rrc_beta - 0.35) < 1e-7 and abs(cfg.agc_rate - 0.02) < 1e-8
    assert abs(cfg.costas_bandwidth - 0.01) < 1e-8 and abs(cfg.fll_bandwidth - 0.006) < 1e-8
    assert abs(cfg.mu_gain - 0.0176028) < 1e-7 and abs(cfg.omega_gain - 1.56359e-4) < 1e-9
    assert abs(cfg.omega_rel_limit - 0.02) < 1e-8


def test_struct_layouts_match_header(pkg):
    """ctypes mirrors of the two ABI structs have the size the C compiler gives them."""
    import subprocess
    import tempfile
    prog = '#include <stdio.h>\n#include "tetra_demod.h"\nint main(){printf("%zu %zu\\n", sizeof(tetra_demod_config_t), sizeof(tetra_demod_channel_state_t));return 0;}\n'
    with tempfile.TemporaryDirectory() as td:
        cfile = os.path.join(td, "s.c")
        open(cfile, "w").write(prog)
        exe = os.path.join(td, "s")
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), cfile, "-o", exe], check=True)
        a, b = map(int, subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split())
    assert a == C.sizeof(pkg.binding.Config) and b == C.sizeof(pkg.binding.ChannelState)


def test_bits_stride_contract(pkg):
    for n in (0, 1, 100, 36000, 1000000):
        s = pkg.binding.bits_stride(n)
        assert s % 16 == 0 and s >= 2 * (n / 1.94 + 1)
    assert pkg.load_library().tetra_demod_bits_stride(-1) < 0


def test_no_cpu_fallback(pkg):
    """Without a GPU the product path must fail loudly, not compute on the CPU."""
    if pkg.binding.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(pkg.TetraDemodError) as e:
        pkg.Demodulator(4, 1024)
    assert e.value.status == -3
    L = pkg.load_library()
    assert b"no usable HIP device" in L.tetra_demod_strerror(-3)
    # every status of include/tetra_demod.h has its own text (TETRA_ERR_OVERRUN = -8 was missing until round 3)
    texts = [L.tetra_demod_strerror(st) for st in range(0, -9, -1)]
    assert len(set(texts)) == 9 and b"unknown" not in b" ".join(texts) and b"unknown" in L.tetra_demod_strerror(-9)
    src = open(os.path.join(ROOT, "sdrpp-tetra-demodulator_amd", "binding.py")).read()
    assert "oracle" not in src.replace("oracle's", "")


def test_product_sources_do_not_reference_oracle():
    pk = os.path.join(ROOT, "sdrpp-tetra-demodulator_amd")
    for dp, _, fs in os.walk(pk):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "tetra_oracle" not in txt.replace("tetra_oracle.c", "").replace("tetra_oracle_sincosf", ""), f
                assert "from oracle" not in txt and "import oracle" not in txt, f


def test_public_headers_are_plain_c99_and_cxx11():
    """Every header under include/ must be consumable by a C compiler (the drop-in boundary is a C ABI) and by C++11."""
    import glob
    import subprocess
    for h in sorted(glob.glob(os.path.join(ROOT, "include", "*.h"))):
        for lang, std in (("c", "-std=c99"), ("c++", "-std=c++11")):
            subprocess.run(["gcc" if lang == "c" else "g++", std, "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", lang, h],
                           check=True)


def test_c_program_links_against_the_library_and_runs_host_only_entry_points(pkg, tmp_path):
    """A C translation unit, compiled by gcc, linked against libtetra_demod_hip.so: the entry points that need no GPU work."""
    import subprocess
    lib = pkg.load_library()._name
    src = tmp_path / "link.c"
    src.write_text('''
#include <stdio.h>
#include "tetra_demod.h"
#include "tetra_chan.h"
#include "tetra_burst_scan.h"
#include "tetra_lmac.h"
#include "tetra_burst_sync.h"
int main(void) {
    tetra_demod_config_t cfg; tetra_chan_config_t cc; tetra_lmac_blk_param_t bp;
    if (tetra_demod_default_config(&cfg) != TETRA_OK || tetra_chan_default_config(&cc) != TETRA_OK) return 1;
    if (tetra_lmac_blk_param(TETRA_TPSAP_T_SCH_F, &bp) != TETRA_OK || bp.type345_bits != 432 || bp.type1_bits != 268) return 2;
    if (tetra_lmac_scramb_init(262, 1, 5) != 0x41800117u) return 3;
    if (tetra_demod_bits_stride(36000) < 36000) return 4;
    if (tetra_bsync_max_frames(NULL) != TETRA_ERR_ARG) return 5;
    /* the ABI-2 additions reject bad arguments before touching the device */
    if (tetra_demod_process_async(NULL, &cfg, TETRA_IQ_CF32, 16, (uint8_t*)&cfg, 32, (int32_t*)&cfg) != TETRA_ERR_ARG) return 6;
    if (tetra_demod_wait(NULL) != TETRA_ERR_ARG) return 7;
    if (tetra_burst_demux_compact_device(NULL, NULL, 4, TETRA_TPSAP_T_SB1, 1, NULL, 120, NULL, NULL, NULL) != TETRA_ERR_ARG) return 8;
    if (tetra_lmac_decode_counted_device(7, NULL, 4, NULL, 120, NULL, NULL, NULL, 80, NULL, NULL) != TETRA_ERR_ARG) return 9;
    { int clk = 0, cus = 0; const int rc = tetra_demod_device_info(12345, &clk, &cus); if (rc != TETRA_ERR_NO_DEVICE) return 10; }
    { tetra_ts_indicator_t* ind = (tetra_ts_indicator_t*)&cfg;
      if (tetra_ts_indicator_create(0, -1, &ind) != TETRA_ERR_ARG || ind != NULL) return 11;
      if (tetra_ts_indicator_create(4, -1, NULL) != TETRA_ERR_ARG) return 12;
      if (tetra_ts_indicator_reset(NULL, -1) != TETRA_ERR_ARG) return 13;
      if (tetra_ts_indicator_process_device(NULL, NULL, 64, NULL, NULL, NULL, NULL) != TETRA_ERR_ARG) return 14;
      tetra_ts_indicator_destroy(NULL); }
    /* the ABI-3 additions likewise */
    { long long ov = 0;
      if (tetra_demod_bits_stride_for(NULL, 16) != TETRA_ERR_ARG || tetra_demod_get_overruns(NULL, &ov) != TETRA_ERR_ARG) return 15;
      if (tetra_demod_set_rrc_params(NULL, 65, 0.35) != TETRA_ERR_ARG) return 16;
      if (tetra_demod_process_resident(NULL, NULL, 0, NULL, 0, NULL, NULL) != TETRA_ERR_ARG) return 17;
      if (tetra_demod_kernel_ms_history(NULL, 1, NULL) != TETRA_ERR_ARG) return 18;
      cfg.flags = TETRA_FLAG_WIDE_WORKGROUPS | TETRA_FLAG_NARROW_WORKGROUPS;       /* contradictory: refused before any device work */
      { tetra_demod_t* hh = NULL; if (tetra_demod_create(&cfg, &hh) != TETRA_ERR_ARG || hh != NULL) return 19; } }
    tetra_demod_host_free(NULL);
    /* ABI 4: the library says what sources it was built from */
    { const char* id = tetra_demod_build_id(); int i; if (!id) return 20; for (i = 0; i < 64; i++) if (!id[i]) return 21; if (id[64]) return 22; }
    /* ABI 5: caller-designed tables on a live handle */
    { float t[4] = { 0 }; if (tetra_demod_set_tables(NULL, t, 2, NULL, 0, NULL) != TETRA_ERR_ARG) return 23; }
    /* ABI 6: the constellation tap */
    { float z[2]; int32_t nb; if (tetra_demod_get_constellation(NULL, 0, 1, z, &nb) != TETRA_ERR_ARG) return 24;
      if (TETRA_CONSTELLATION_SYMBOLS != 1024 || TETRA_FLAG_CONSTELLATION != 256) return 25; }
    /* round 6: frame lists, decoding straight from packed frames, the list tracker, integer captures, the chain handle -- arguments first */
    { tetra_lmac_frames_t fr = { 0 }; tetra_lmac_job_t job = { 0 };
      if (tetra_burst_index_device(NULL, 8, 4, NULL, NULL, NULL, NULL, NULL) != TETRA_ERR_ARG) return 26;
      if (tetra_lmac_decode_frames_device(NULL, &job, 1, NULL) != TETRA_ERR_ARG || tetra_lmac_decode_frames_device(&fr, &job, 9, NULL) != TETRA_ERR_ARG) return 27;
      if (tetra_lmac_decode_frames_device(&fr, &job, 0, NULL) != TETRA_OK) return 28;
      job.type = TETRA_TPSAP_T_SCH_F; job.max_rows = 640;
      if (tetra_lmac_decode_frames_workspace_bytes(&job, 1) != (size_t)10 * 146 * 64 * 4 || tetra_lmac_decode_frames_workspace_bytes(NULL, 1) != 0) return 29;
      if (tetra_lmac_track_sync_lists_device(NULL, 80, NULL, NULL, NULL, NULL, 1, 4, NULL, NULL, NULL, NULL, NULL, NULL, NULL) != TETRA_ERR_ARG) return 30;
      if (tetra_chan_process_device_cs16(NULL, NULL, 0, NULL, NULL, NULL) != TETRA_ERR_ARG || tetra_chan_process_device_cs8(NULL, NULL, 0, NULL, NULL, NULL) != TETRA_ERR_ARG) return 31;
      if (sizeof(tetra_lmac_label_t) != 24 || TETRA_N_LISTS != 4 || TETRA_LMAC_MAX_JOBS != 8) return 32; }
    printf("%d %d %d\\n", (int)cfg.rrc_tap_count, (int)cc.n_channels, tetra_demod_abi_version());
    return 0;
}
''')
    exe = tmp_path / "link"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-I", os.path.join(ROOT, "include"), str(src), lib, "-Wl,-rpath," + os.path.dirname(lib),
                    "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert out == ["65", "800", "6"]


def test_generated_fll_assembly_is_current():
    """csrc/fll_asm.inc is generated by csrc/gen_fll_asm.py and committed: the committed file is what the generator emits."""
    import subprocess
    import sys
    gen = os.path.join(ROOT, "sdrpp-tetra-demodulator_amd", "csrc", "gen_fll_asm.py")
    r = subprocess.run([sys.executable, gen, "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
