"""CPU: the C-ABI shared library loads and exports every symbol that
include/loudscan.h, include/loudscan_device.h and include/loudscan_ebur128.h declare
(no compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b((?:scan|lgd|ebur128|loudscan_ebur128)_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from loudgain_amd import _lib, ebur128, scan
    L = _lib.load()
    dev = _declared("loudscan_device.h")
    api = _declared("loudscan.h")
    shim = _declared("loudscan_ebur128.h")
    assert len(dev) >= 15 and len(api) >= 14 and len(shim) >= 12
    for name in dev + api + shim:
        assert hasattr(L, name), "missing export: " + name
    assert sorted(_lib.DEVICE_SYMBOLS) == dev
    assert sorted(scan.SCAN_SYMBOLS) == api
    assert sorted(ebur128.EBUR128_SYMBOLS) == shim


def test_ebur128_state_layout_and_version():
    # libebur128's public struct: int mode; unsigned channels; unsigned long samplerate; d* (LP64: 24 bytes);
    # loudgain.c:179-184 insists on >= 1.2.4
    from loudgain_amd import ebur128
    assert ctypes.sizeof(ebur128.Ebur128State) == 24
    assert ebur128.Ebur128State.channels.offset == 4 and ebur128.Ebur128State.samplerate.offset == 8
    assert ebur128.get_version() >= (1, 2, 4)
    assert ebur128.MODE_ALL == 0b111111 and ebur128.MODE_TRUE_PEAK == 0x31 and ebur128.MODE_LRA == 0xB


def test_ebur128_init_fails_without_gpu():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from loudgain_amd import ebur128
    with pytest.raises(ebur128.Ebur128Error):
        ebur128.State(2, 48000)


def test_scan_result_layout_matches_reference_header():
    # /root/reference/src/scan.h:35-53 on LP64: 2 pointers, int (+pad), 9 doubles = 96 bytes
    from loudgain_amd.scan import ScanResult
    assert ctypes.sizeof(ScanResult) == 96
    names = [f[0] for f in ScanResult._fields_]
    assert names == ["file", "container", "codec_id", "track_gain", "track_peak", "track_loudness",
                     "track_loudness_range", "album_gain", "album_peak", "album_loudness",
                     "album_loudness_range", "loudness_reference"]
    assert ScanResult.track_gain.offset == 24 and ScanResult.loudness_reference.offset == 88


def test_no_cpu_fallback_without_gpu():
    """Without a GPU the product refuses to run instead of computing on the CPU."""
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from loudgain_amd.device import DeviceScanner, LoudscanError
    with pytest.raises(LoudscanError):
        DeviceScanner(0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "loudgain_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "lgoracle" not in txt and "liblgoracle" not in txt and "lg_oracle" not in txt, f
