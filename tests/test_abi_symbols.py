"""The C-ABI library loads on a CPU-only machine and exports every symbol include/heat_amd.h declares.
No compute calls here (those need a GPU and live in test_parity_gpu.py)."""
import ctypes
import os
import re

import numpy as np
import pytest

from heat_amd import binding, modeldict as mdl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = set()
    for h in ("heat_amd.h", "heat_amd_setup.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(heat_[a-z_0-9]+)\s*\(", text))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    from heat_amd import build as hb
    hb.build()
    lib = ctypes.CDLL(binding.lib_path())
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "libheat_amd.so does not export %s" % n


def test_binding_covers_the_header():
    bound = {n for n, _, _ in binding.SYMBOLS}
    assert bound == set(declared_functions())
    binding.load_library()
    assert binding.load_library().heat_amd_abi_version() == 1


def test_struct_layouts_match_the_header(tmp_path):
    # the ctypes mirrors must have the size and field offsets gcc gives the header's structs
    import subprocess
    src = tmp_path / "sz.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "heat_amd.h"\n'
        'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(heat_cavity), sizeof(heat_weather), '
        'sizeof(heat_batch_options), sizeof(heat_batch_desc), offsetof(heat_batch_desc, node_offset), '
        'offsetof(heat_batch_desc, zone_slot), offsetof(heat_batch_options, n_ranks));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert got == [ctypes.sizeof(binding.Cavity), ctypes.sizeof(binding.Weather), ctypes.sizeof(binding.Options),
                   ctypes.sizeof(binding.Desc), binding.Desc.node_offset.offset, binding.Desc.zone_slot.offset,
                   binding.Options.n_ranks.offset]


def test_missing_gpu_fails_loudly():
    import subprocess, sys
    # On a machine without a HIP device the product must refuse to compute (no CPU fallback).
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from heat_amd import HeatBatch, HeatError, modeldict as mdl\n"
        "md, st = mdl.uniform_massive(10, 8, Z=1)\n"
        "try:\n"
        "    HeatBatch(md)\n"
        "    print('CREATED')\n"
        "except HeatError as e:\n"
        "    print('ERR', e.code)\n" % ROOT)
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120).stdout
    assert "ERR -5" in out, out  # HEAT_E_DEVICE


def test_descriptor_validation_messages():
    md, st = mdl.uniform_massive(4, 8, Z=1)
    d, keep = binding.make_desc(md)
    assert d.n_surfaces == 4 and d.n_state == md["n_state"]
    assert keep["node_offset"][-1] == 32
