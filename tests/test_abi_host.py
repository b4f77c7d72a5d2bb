"""The C-ABI library loads and exports every symbol include/*.h declares; host-only entry points (packer,
QV thresholds from the environment, error model, synthetic host generator) behave.  No GPU compute."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import cases
from minorseq_amd import abi, device, synth

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def header_functions():
    names = set()
    for h in ("msq_abi.h", "msq_synth.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(msq_[a-z0-9_]+)\s*\(", text))
    return names


def test_every_declared_symbol_is_exported():
    L = abi.lib()
    declared = header_functions()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(L, name), name
    assert declared == set(abi.declared_symbols())


def test_pack_bit_planes():
    reads = cases.make_reads([(7, "=XDI" + "=" * 30 + "PS", "ACG-T"[0:0] + "AC-T" + "G" * 30 + "*A"), (3, "==", "TN")])
    p = device.pack_host(reads)
    assert p["n_reads"] == 2 and list(p["group_off"]) == [0, 2, 3] and p["n_groups"] == 3
    bop = p["bop"].reshape(-1, 4)
    # normalized groups (MSQ_READS_NORMALIZED): consuming bases from slot 0 up, the others from slot 31 down
    # read 0, group 0: A C - (I:T -> slot 31) then 28 G
    assert bop[0, 0] & 0x7 == 0b010 and bop[0, 1] & 0x7 == 0b000         # A=0 C=1 (D code 0)
    assert bop[0, 2] & 0x7 == 0b100 and bop[0, 3] & 0x7 == 0b000         # ops 0 0 1
    assert bop[0, 0] >> 3 == 0x10000000 and bop[0, 1] >> 3 == 0x1FFFFFFF   # 28 G = 2, then the inserted T = 3
    assert bop[0, 2] >> 3 == 0 and bop[0, 3] >> 3 == 0x10000000            # ... op 2 in slot 31
    # group 1: 2 G, then padding, S in slot 30 and P in slot 31 (all op 3)
    assert bop[1, 2] == 0xFFFFFFFC and bop[1, 3] == 0xFFFFFFFC
    assert bop[1, 0] == 0 and bop[1, 1] == 0b11
    # read 1: T then literal N -> op 1 code 1
    assert bop[2, 0] & 3 == 0b11 and bop[2, 1] & 3 == 0b01 and bop[2, 2] & 3 == 0b10 and bop[2, 3] & 3 == 0
    assert p["flags"] == abi.MSQ_READS_NORMALIZED
    q = p["qual"].reshape(-1, 32)
    qv_byte = lambda k: 4 * (k & 7) + (k >> 3)  # MSQ_QV_BYTE
    assert all(q[0, qv_byte(k)] == 60 for k in range(32))
    assert [int(q[1, qv_byte(k)]) for k in range(32)] == [60] * 2 + [0] * 28 + [60] * 2
    assert p["subqv"] is None


def test_pack_errors_and_exotic():
    bad = cases.make_reads([(0, "==N", "ACG")])
    with pytest.raises(abi.MsqError) as e:
        device.pack_host(bad)
    assert e.value.code == abi.MSQ_ERR_CIGAR and "Unexpected cigar 78" in str(e.value)  # std::to_string('N')
    ex = device.pack_host(cases.make_reads([(0, "===", "ARG")]))
    assert ex["flags"] & abi.MSQ_READS_HAS_EXOTIC and ex["flags"] & abi.MSQ_READS_NORMALIZED
    empty = device.pack_host(cases.make_reads([]))
    assert empty["n_reads"] == 0 and empty["n_groups"] == 0


def test_qv_thresholds_from_env(monkeypatch):
    L = abi.lib()
    for k in ("DELQV", "SUBQV", "INSQV", "QUALQV"):
        monkeypatch.delenv(k, raising=False)
    t = abi.QvThresholds()
    abi.check(L.msq_qv_thresholds_from_env(C.byref(t)))
    assert (t.qual, t.del_, t.sub, t.ins) == (-1, -1, 42, -1)  # src/QvThresholds.cpp:52-55
    monkeypatch.setenv("QUALQV", "20")
    monkeypatch.setenv("SUBQV", "300")  # stoi -> uint8_t narrowing
    abi.check(L.msq_qv_thresholds_from_env(C.byref(t)))
    assert (t.qual, t.sub) == (20, 300 & 255)
    monkeypatch.setenv("DELQV", "abc")
    assert L.msq_qv_thresholds_from_env(C.byref(t)) == abi.MSQ_ERR_ARG


def test_error_model():
    L = abi.lib()
    em = abi.ErrorModel()
    abi.check(L.msq_error_model_default(C.byref(em)))
    assert (em.match, em.substitution, em.deletion) == (0.9956844883, 0.0005244257 / 3.0, 0.003791086)
    abi.check(L.msq_error_model_from_rates(0.001, 0.005, C.byref(em)))
    assert (em.match, em.substitution, em.deletion) == (1 - 0.001 - 0.005, 0.001 / 3.0, 0.005)


def test_synth_host_is_deterministic_and_shardable():
    wl = synth.workload("pol_100k", n_reads=64)
    a = wl.host_reads(0, 64)
    b = wl.host_reads(32, 32)
    off = a["base_off"][32]
    assert (a["cigar"][off:] == b["cigar"]).all() and (a["nuc"][off:] == b["nuc"]).all()
    assert (a["qual"][off:] == b["qual"]).all()
    assert (a["ref_start"] == 2252).all() and (a["ref_end"] == 5095).all()
    cig = bytes(a["cigar"]).decode()
    n = len(cig)
    assert 0.002 < cig.count("D") / n < 0.006 and 0.0003 < cig.count("I") / n < 0.002 and cig.count("X") > 0
    assert len(wl.minor_sites) == 20
    rich = synth.workload("hiv_1k", n_reads=8, rich_qv=True).host_reads(0, 8)
    assert rich["subqv"] is not None and (rich["ref_start"] >= 0).all()


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    assert abi.lib().msq_ctx_create(0, C.byref(h)) == abi.MSQ_ERR_NODEVICE
    assert b"no CPU fallback" in abi.lib().msq_last_error()
