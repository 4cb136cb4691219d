"""Plan-specialised kernels: compile ``maro_amd/csrc/cim_spec.hip`` (CIM) or ``cb_spec.hip`` (citi_bike) for one plan with every
integer dimension / layout offset of the plan as a compile-time constant (the text of ``mrx_cim_plan_defines`` /
``mrx_cb_plan_defines``), cache the gfx950 code object in-tree (``maro_amd/csrc/spec_cache/`` — git-ignored like the built
``.so``, so it travels with a repo snapshot), and hand it to ``mrx_cim_load_step_kernels`` / ``mrx_cb_load_step_kernels``.  Compiling needs ``hipcc`` (no GPU); a few seconds per plan, once.
"""
from __future__ import annotations

import ctypes
import hashlib
import os
import shutil
import subprocess
import tempfile

from .. import _lib

CSRC = os.environ.get("MARO_AMD_CSRC") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc")   # ($MARO_AMD_CSRC: A/B experiments against another revision's device sources)
CACHE = os.environ.get("MARO_AMD_SPEC_CACHE", os.path.join(CSRC, "spec_cache"))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--genco", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value"] + os.environ.get("MARO_AMD_SPEC_FLAGS", "").split()
# the device code's results are bit-exact against the reference because fp64 arithmetic is compiled as written: flags that let
# the compiler re-associate, contract or approximate it are refused outright (they would also change the code-object key)
_UNSAFE = ("-ffast-math", "-Ofast", "-funsafe-math-optimizations", "-freciprocal-math", "-fassociative-math", "-fno-signed-zeros", "-ffp-contract=fast",
           "-fapprox-func", "-ffinite-math-only", "-ffp-model=fast", "-ffp-model=aggressive")
if any(f in _UNSAFE for f in FLAGS):
    raise ValueError(f"MARO_AMD_SPEC_FLAGS: {[f for f in FLAGS if f in _UNSAFE]} would break bit-exact fp64 (order generator, transfer times)")
UNITS = {   # scenario -> (translation unit, generated dims header, the sources the cache key covers, ABI prefix)
    "cim": ("cim_spec.hip", "cim_spec_dims.h", ("cim_spec.hip", "cim_step_kernels.h", "cim_device.h", "cim_params.h", "cim_prof.h", "wave.h"), "mrx_cim"),
    "citi_bike": ("cb_spec.hip", "cb_spec_dims.h", ("cb_spec.hip", "cb_step_kernels.h", "cb_device.h", "cb_wave.h", "cb_params.h", "wave.h",
                                                    "../../include/maro_amd_citi_bike.h"), "mrx_cb"),
}


def plan_defines(topo_struct, cfg, scenario: str = "cim", obs=((), ())) -> str:
    """The plan's ``#define MRXC_<field> <value>`` text (host only; no device needed).  CIM: `obs` = (port attribute ids,
    vessel attribute ids) of the fused observation (mrx_cim_set_observation), part of what is compiled in."""
    fn = getattr(_lib.load(), UNITS[scenario][3] + "_plan_defines")
    extra = ()
    if scenario == "cim":
        pa, va = ((ctypes.c_int32 * len(x))(*x) for x in obs)
        extra = (pa, len(obs[0]), va, len(obs[1]))
    n = _lib.check(fn(ctypes.byref(topo_struct), ctypes.byref(cfg), *extra, None, 0), fn.__name__)
    buf = ctypes.create_string_buffer(n)
    _lib.check(fn(ctypes.byref(topo_struct), ctypes.byref(cfg), *extra, buf, n), fn.__name__)
    return buf.value.decode()


_TOOLCHAIN = None


def _toolchain() -> bytes:
    """The compiler's identity, part of the cache key: a ROCm upgrade must not load code objects of the old compiler.  Where
    hipcc is absent (a box that only consumes the cache) the version recorded by the last build is used."""
    global _TOOLCHAIN
    if _TOOLCHAIN is None:
        stamp = os.path.join(CACHE, "TOOLCHAIN")
        hipcc = HIPCC if os.path.exists(HIPCC) else shutil.which("hipcc")
        try:
            _TOOLCHAIN = subprocess.run([hipcc, "--version"], capture_output=True, check=True).stdout if hipcc else None
        except (OSError, subprocess.CalledProcessError):
            _TOOLCHAIN = None
        if _TOOLCHAIN is None:
            _TOOLCHAIN = open(stamp, "rb").read() if os.path.exists(stamp) else b"unknown"
        else:
            os.makedirs(CACHE, exist_ok=True)
            if not os.path.exists(stamp) or open(stamp, "rb").read() != _TOOLCHAIN:
                with open(stamp, "wb") as f:
                    f.write(_TOOLCHAIN)
    return _TOOLCHAIN


def _key(defines: str, scenario: str) -> str:
    h = hashlib.sha256(defines.encode())
    for name in UNITS[scenario][2]:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    h.update(_toolchain())
    return h.hexdigest()[:24]


def code_object(defines: str, build: bool = True, scenario: str = "cim") -> bytes:
    """The gfx950 code object of the scenario's spec unit for `defines` — from the cache, else compiled now (build=False: KeyError)."""
    unit, dims_header = UNITS[scenario][:2]
    path = os.path.join(CACHE, _key(defines, scenario) + ".hsaco")
    if not os.path.exists(path):
        if not build:
            raise KeyError(path)
        hipcc = HIPCC if os.path.exists(HIPCC) else shutil.which("hipcc")
        if not hipcc:
            raise RuntimeError("specialised CIM kernels requested but hipcc was not found (set HIPCC, or use specialize=False)")
        os.makedirs(CACHE, exist_ok=True)
        with tempfile.TemporaryDirectory() as tmp:
            with open(os.path.join(tmp, dims_header), "w") as f:
                f.write("// generated: " + UNITS[scenario][3] + "_plan_defines\n" + defines)
            out = os.path.join(tmp, "spec.hsaco")
            subprocess.check_call([hipcc] + FLAGS + ["-I", tmp, "-I", CSRC, "-o", out, os.path.join(CSRC, unit)])
            tmp_dst = path + f".{os.getpid()}.tmp"
            shutil.copyfile(out, tmp_dst)
            os.replace(tmp_dst, path)   # atomic: several ranks may build the same plan concurrently
    with open(path, "rb") as f:
        return f.read()


LOADS = 0   # code objects handed to engines in this process (diagnostics: e.g. that a sweep really ran specialised)


def load_into(engine, defines: str, build: bool = True, scenario: str = "cim") -> None:
    global LOADS
    img = code_object(defines, build, scenario)
    buf = ctypes.create_string_buffer(img, len(img))
    fn = getattr(_lib.load(), UNITS[scenario][3] + "_load_step_kernels")
    _lib.check(fn(engine._h, buf, len(img), defines.encode()), fn.__name__)
    # identity of what is now running: the cache key (plan text + every source the unit includes + flags + toolchain) and the
    # image's own hash — bench.py stamps its line with it, and a PMC record is only used for a run of the SAME code object
    engine.code_object_key = _key(defines, scenario)
    engine.code_object_sha16 = hashlib.sha256(img).hexdigest()[:16]
    LOADS += 1
