"""The C-ABI library loads and exports every symbol include/ipx.h declares (no GPU needed), and the
host-only entry points (geometry, parsing) answer the committed known answers."""
import json
import os
import re

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def ipx():
    from imageprocessor_amd import build
    build.build()
    import imageprocessor_amd as m
    return m


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "ipx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ipx_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(ipx):
    L = ipx.lib()
    names = _header_symbols()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_ctypes_table_matches_header(ipx):
    from imageprocessor_amd import _lib
    assert sorted(_lib.SIGNATURES) == _header_symbols()


def test_abi_version_and_error_text(ipx):
    L = ipx.lib()
    assert L.ipx_abi_version() == 1
    with pytest.raises(ipx.IpxError) as e:
        ipx.resize_dims(100, 100, 0, 5, True)
    assert e.value.status == -1 and "positive" in e.value.text   # resize.go:51-53


def test_no_gpu_is_an_error_not_a_fallback(ipx):
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present")
    assert ipx.device_count() < 0
    with pytest.raises(ipx.IpxError) as e:
        ipx.Context()
    assert e.value.status == -5 and "no CPU fallback" in e.value.text


def test_product_does_not_touch_the_oracle():
    """Nothing under imageprocessor_amd/ may import, link or name the oracle."""
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "imageprocessor_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(d, f), errors="replace").read()
                if re.search(r"\boracle\b|ipxo_|ipx_oracle", txt):
                    bad.append(f)
    assert not bad, bad


def test_host_rules_against_golden(ipx):
    with open(os.path.join(HERE, "golden", "kats.json")) as f:
        cases = json.load(f)["cases"]
    n = 0
    for c in cases:
        k = c["kind"]
        if k == "resize_dims":
            assert list(ipx.resize_dims(c["ow"], c["oh"], c["w"], c["h"], c["keep_aspect"])) == c["expect"]
        elif k == "thumb_geometry":
            crop, nw, nh = ipx.thumb_geometry(c["ow"], c["oh"], c["size"], c["crop_to_fit"])
            assert list(crop) == c["expect"]["crop"] and (nw, nh) == (c["expect"]["nw"], c["expect"]["nh"])
        elif k == "text_height":
            assert ipx.text_height_px(c["font_size"]) == c["expect"]
        elif k == "anchor":
            assert list(ipx.watermark_anchor(c["position"], c["w"], c["h"], c["width_px"], c["height_px"])) == c["expect"]
        elif k == "parse_color":
            rgba, err = ipx.parse_color(c["s"], c["opacity"])
            assert list(rgba) == c["expect"]["rgba"] and err == c["expect"]["error"], c
        else:
            continue
        n += 1
    assert n >= 30


def test_host_rules_agree_with_oracle_on_a_sweep(ipx):
    import oracle
    import random
    rng = random.Random(5)
    for _ in range(2000):
        ow, oh = rng.randrange(1, 9000), rng.randrange(1, 9000)
        w, h = rng.randrange(1, 3000), rng.randrange(1, 3000)
        keep = rng.random() < 0.5
        assert ipx.resize_dims(ow, oh, w, h, keep) == oracle.resize_dims(ow, oh, w, h, keep)
        size, crop = rng.randrange(1, 600), rng.random() < 0.5
        assert ipx.thumb_geometry(ow, oh, size, crop) == oracle.thumb_geometry(ow, oh, size, crop)
    for fs in (1, 7.5, 12, 36, 36.4, 72, 100.01):
        assert ipx.text_height_px(fs) == oracle.text_height_px(fs)
    for s in ("255,255,255", " 1, 2,3 ,4", "a,b,c", "1,2,3,", "-1,999,5", "+7,8,9", "1,,3", "1,2,3,4,5", "0x10,1,1"):
        for op in (0.0, 0.3, 0.5, 1.0):
            assert ipx.parse_color(s, op) == oracle.parse_color(s, op), (s, op)


def test_frame_geometry_guard(ipx):
    """Frames the 32-bit buffer descriptors cannot address are refused with IPX_ERR_UNSUPPORTED (the worker keeps its CPU path,
    domain/task.go:55: a 32 MiB PNG can decode past 23170 x 23170), never processed with wrapped offsets."""
    L = ipx.lib()
    ok = [(1920, 1080, 7680, 4), (7680, 4320, 30720, 4), (23168, 23168, 92672, 4), (32764, 16380, 131056, 4), (65535, 8000, 65535, 1),
          (0, 0, 0, 4), (65535, 1, 262140, 4)]
    for w, h, stride, bpp in ok:
        assert L.ipx_frame_supported(w, h, stride, bpp) == 0, (w, h, stride)
    unsupported = [(23171, 23171, 92684, 4),      # 2^31 bytes and a bit
                   (32768, 16384, 131072, 4),    # exactly 2 GiB
                   (65536, 16, 262144, 4), (16, 65536, 64, 4),   # a side beyond 65535
                   (1920, 1080, 2 << 20, 4)]     # a small frame inside an enormous pitch
    for w, h, stride, bpp in unsupported:
        assert L.ipx_frame_supported(w, h, stride, bpp) == -4, (w, h, stride)
        assert b"span" in L.ipx_last_error()
    for w, h, stride, bpp in [(-1, 5, 100, 4), (10, 10, 39, 4), (10, 10, 40, 0)]:
        assert L.ipx_frame_supported(w, h, stride, bpp) == -1
    # the largest tight RGBA8 frame whose last byte a 31-bit offset reaches
    assert L.ipx_frame_supported(23169, 23169, 23169 * 4, 4) == 0
    assert 23169 * 23169 * 4 < 0x7fff0000 <= 23170 * 23170 * 4 + 0x10000
