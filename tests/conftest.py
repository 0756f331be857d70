import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import libeddsa_amd  # noqa: E402
# the suite drives the hooks of include/eddsa_amd_debug.h (route selection, fault injectors, counters): it binds the debug build -
# the product's object files plus those functions.  The shipped library is what the C programs under tests/c link and what
# test_abi_and_host.py inspects.
libeddsa_amd.use_debug_library()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class Oracle:
    """ctypes view of oracle/liboracle.so (the CPU restatement; test infrastructure only)."""

    def __init__(self, lib):
        self.lib = lib
        self.threads = max(1, min(os.cpu_count() or 1, 16))

    @staticmethod
    def _p(a):
        return a.ctypes.data_as(ctypes.c_void_p)

    def _one(self, name, out_len, *args):
        out = ctypes.create_string_buffer(out_len)
        getattr(self.lib, name)(out, *args)
        return out.raw

    def genpub(self, sk): return self._one("orc_ed25519_genpub", 32, sk)
    def sign(self, sk, pk, m): return self._one("orc_ed25519_sign", 64, sk, pk, m, ctypes.c_size_t(len(m)))
    def verify(self, sig, pk, m): return bool(self.lib.orc_ed25519_verify(sig, pk, m, ctypes.c_size_t(len(m))))
    def x25519(self, s, p): return self._one("orc_x25519", 32, s, p)
    def x25519_base(self, s): return self._one("orc_x25519_base", 32, s)
    def pk_to_x(self, p): return self._one("orc_pk_ed25519_to_x25519", 32, p)
    def sk_to_x(self, s): return self._one("orc_sk_ed25519_to_x25519", 32, s)
    def sha512(self, m): return self._one("orc_sha512", 64, m, ctypes.c_size_t(len(m)))

    def verify_batch(self, sig, pk, msg, msg_len):
        n = sig.shape[0]
        ok = np.zeros(n, np.uint8)
        self.lib.orc_ed25519_verify_batch(self._p(ok), self._p(sig), self._p(pk), self._p(msg),
                                          ctypes.c_size_t(msg_len), ctypes.c_size_t(n), self.threads)
        return ok

    def x25519_batch(self, sc, pt):
        n = sc.shape[0]
        out = np.zeros((n, 32), np.uint8)
        self.lib.orc_x25519_batch(self._p(out), self._p(sc), self._p(pt), ctypes.c_size_t(n), self.threads)
        return out

    def sign_batch(self, sk, pk, msg, msg_len):
        n = sk.shape[0]
        out = np.zeros((n, 64), np.uint8)
        self.lib.orc_ed25519_sign_batch(self._p(out), self._p(sk), self._p(pk), self._p(msg),
                                        ctypes.c_size_t(msg_len), ctypes.c_size_t(n), self.threads)
        return out

    def genpub_batch(self, sk):
        n = sk.shape[0]
        out = np.zeros((n, 32), np.uint8)
        self.lib.orc_ed25519_genpub_batch(self._p(out), self._p(sk), ctypes.c_size_t(n), self.threads)
        return out


@pytest.fixture(scope="session")
def oracle():
    path = os.path.join(ROOT, "oracle", "liboracle.so")
    src = os.path.join(ROOT, "oracle", "eddsa_oracle.c")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
    return Oracle(ctypes.CDLL(path))


@pytest.fixture(scope="session")
def reflib():
    """the compiled reference (oracle/_ref); built here when the reference checkout exists"""
    path = os.path.join(ROOT, "oracle", "_ref", "libeddsa_ref.so")
    if not os.path.exists(path):
        if os.path.isdir("/root/reference/lib"):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
        else:
            pytest.skip("oracle/_ref not built and no reference checkout")
    lib = ctypes.CDLL(path)
    lib.ed25519_verify.restype = ctypes.c_bool
    return lib


@pytest.fixture(scope="session")
def golden():
    def load(name):
        path = os.path.join(GOLD, name)
        if name.endswith(".json"):
            return json.load(open(path))
        return open(path, "rb").read()
    return load


@pytest.fixture(scope="session")
def engine():
    """the product, bound to cuda:0; GPU tests only"""
    import torch
    import libeddsa_amd as ed
    assert torch.cuda.is_available()
    ed.init(0)
    return ed


@pytest.fixture(scope="session")
def hostcheck():
    """the device source compiled for the host CPU with every bound asserted (tests/host_check/):
    a test binary that lets the CPU suite run the kernels' exact algorithms; not the product"""
    d = os.path.join(ROOT, "tests", "host_check")
    lib = os.path.join(d, "libhostcheck.so")
    srcs = [os.path.join(d, "host_check.cpp")] + [os.path.join(ROOT, "libeddsa_amd", "csrc", f)
                                                   for f in ("lanes.h", "halve.h", "rlc_lanes.h", "fe25519.h", "ge25519.h", "sc25519.h", "sha512.h")]
    if not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-DED_HOST_CHECK", "-Wno-unknown-pragmas",
                               "-I" + os.path.join(ROOT, "libeddsa_amd", "csrc"), srcs[0], "-o", lib])
    h = ctypes.CDLL(lib)
    h.hc_first_violation.restype = ctypes.c_char_p
    h.hc_violations.restype = ctypes.c_long
    h.hc_reset()
    return h
