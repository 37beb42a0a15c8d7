#!/usr/bin/env python3
"""Writes tests/golden/ref_codec/: small PNG files of every colour type / bit depth / tRNS / interlace /
filter combination plus damaged files, a JSON document of edge-case numbers, and what the REFERENCE's own
decoder and JSON library make of them (expected.npz).  The expected values come from oracle/_ref/ref_codec,
i.e. /root/reference/lib/lodepng.cpp and lib/json.h compiled unmodified (`make -C oracle ref`), so this script
runs only where /root/reference exists; the committed fixtures travel.

Inputs are produced by the PNG writer below (zlib + struct, nothing from the reference)."""
import json
import os
import struct
import subprocess
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "ref_codec")
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_CODEC = os.path.join(ROOT, "oracle", "_ref", "ref_codec")

CHANNELS = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}
ADAM7 = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]   # x0, y0, dx, dy


def chunk(kind, body, bad_crc=False):
    crc = zlib.crc32(kind + body) & 0xFFFFFFFF
    if bad_crc:
        crc ^= 0x5A5A5A5A
    return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", crc)


def pack_rows(samples, depth):
    """samples: [h][w*channels] integer sample values -> list of packed byte rows."""
    rows = []
    for r in samples:
        if depth == 8:
            rows.append(bytes(int(v) for v in r))
        elif depth == 16:
            rows.append(b"".join(struct.pack(">H", int(v)) for v in r))
        else:
            bits = "".join(format(int(v), f"0{depth}b") for v in r)
            bits += "0" * (-len(bits) % 8)
            rows.append(bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8)))
    return rows


def paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def filter_rows(rows, bpp, first_filter):
    """Apply filter types cycling 0..4 from `first_filter`, so every predictor is exercised."""
    out = bytearray()
    prev = bytes(len(rows[0])) if rows else b""
    for y, row in enumerate(rows):
        ft = (first_filter + y) % 5
        line = bytearray()
        for i, v in enumerate(row):
            a = row[i - bpp] if i >= bpp else 0
            b = prev[i] if len(prev) == len(row) else 0
            c = prev[i - bpp] if (i >= bpp and len(prev) == len(row)) else 0
            pred = [0, a, b, (a + b) >> 1, paeth(a, b, c)][ft]
            line.append((v - pred) & 0xFF)
        out.append(ft); out += line
        prev = row
    return bytes(out)


def encode_png(samples, ctype, depth, plte=None, trns=None, interlace=0, first_filter=0, idat_split=1, bad_crc=None, extra=None):
    """samples: uint array [h][w][channels]."""
    h, w, ch = samples.shape
    assert ch == CHANNELS[ctype]
    bpp = max(1, ch * depth // 8)
    if not interlace:
        raw = filter_rows(pack_rows(samples.reshape(h, w * ch), depth), bpp, first_filter)
    else:
        raw = b""
        for (x0, y0, dx, dy) in ADAM7:
            sub = samples[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]:
                raw += filter_rows(pack_rows(sub.reshape(sub.shape[0], -1), depth), bpp, first_filter)
    comp = zlib.compress(raw, 6)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, interlace), bad_crc == "IHDR")
    if extra and extra[0] == "before":
        png += chunk(extra[1], extra[2])
    if plte is not None:
        png += chunk(b"PLTE", bytes(plte))
    if trns is not None:
        png += chunk(b"tRNS", bytes(trns))
    n = max(1, len(comp) // idat_split)
    parts = [comp[i:i + n] for i in range(0, len(comp), n)]
    for k, part in enumerate(parts):
        png += chunk(b"IDAT", part, bad_crc == "IDAT" and k == 0)
    png += chunk(b"IEND", b"")
    return png


def cases():
    rng = np.random.default_rng(20161127)
    out = {}

    def rnd(h, w, ch, depth):
        return rng.integers(0, 1 << depth, size=(h, w, ch), dtype=np.uint32)

    n = 0
    for ctype, depths in ((0, (1, 2, 4, 8, 16)), (2, (8, 16)), (3, (1, 2, 4, 8)), (4, (8, 16)), (6, (8, 16))):
        for depth in depths:
            for (h, w) in ((7, 13), (1, 1), (9, 5)):
                for interlace in (0, 1):
                    ch = CHANNELS[ctype]
                    s = rnd(h, w, ch, depth)
                    plte = trns = None
                    if ctype == 3:
                        ncol = 1 << depth
                        plte = rng.integers(0, 256, size=3 * ncol, dtype=np.uint8)
                        if (h, w) != (9, 5):
                            trns = rng.integers(0, 256, size=max(1, ncol // 2), dtype=np.uint8)   # shorter than the palette
                    elif ctype in (0, 2) and (h, w) == (7, 13):
                        key = s[3, 4]                                           # an existing pixel becomes transparent
                        trns = b"".join(struct.pack(">H", int(v)) for v in key)
                    name = f"c{ctype}_d{depth}_{w}x{h}_i{interlace}"
                    out[name] = encode_png(s, ctype, depth, plte, trns, interlace, first_filter=n % 5, idat_split=1 + n % 3)
                    n += 1
    # 16-bit key that differs from a pixel only in the low byte: must stay opaque
    s = np.full((2, 3, 1), 0x1234, np.uint32); s[0, 1] = 0x1235
    out["c0_d16_lowbyte_key"] = encode_png(s, 0, 16, trns=struct.pack(">H", 0x1235))
    s = np.full((2, 2, 3), 0x8001, np.uint32); s[1, 1] = (0x8001, 0x8002, 0x8001)
    out["c2_d16_lowbyte_key"] = encode_png(s, 2, 16, trns=struct.pack(">HHH", 0x8001, 0x8002, 0x8001))
    # an ancillary chunk the decoders do not know is skipped
    out["unknown_ancillary"] = encode_png(rnd(3, 4, 3, 8), 2, 8, extra=("before", b"teSt", b"hello"))
    # damaged or unusual files: the reference's verdict (error or pixels) is whatever its decoder says
    s = rnd(4, 4, 1, 4)
    out["bad_palette_index"] = encode_png(np.full((4, 4, 1), 9, np.uint32), 3, 4, plte=rng.integers(0, 256, 3 * 4, dtype=np.uint8))
    out["bad_crc_idat"] = encode_png(rnd(4, 4, 4, 8), 6, 8, bad_crc="IDAT")
    out["bad_crc_ihdr"] = encode_png(rnd(4, 4, 4, 8), 6, 8, bad_crc="IHDR")
    out["unknown_critical"] = encode_png(rnd(3, 4, 3, 8), 2, 8, extra=("before", b"TeSt", b"hello"))
    good = encode_png(rnd(6, 6, 4, 8), 6, 8)
    out["truncated"] = good[:len(good) - 30]
    out["not_png"] = b"GIF89a" + good[6:]
    out["bad_depth_rgb4"] = good[:24] + bytes([4, 2]) + good[26:]             # IHDR says RGB at 4 bits (CRC now wrong too)
    s = rnd(5, 5, 3, 8)
    flt = bytearray(filter_rows(pack_rows(s.reshape(5, 15), 8), 3, 0)); flt[16] = 7            # filter type 7 on row 1
    comp = zlib.compress(bytes(flt))
    out["bad_filter_type"] = (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 5, 5, 8, 2, 0, 0, 0)) + chunk(b"IDAT", comp) + chunk(b"IEND", b""))

    # headers that announce more pixels than a decoder may size its buffers for (valid CRCs, a few bytes of image data):
    # a 32-bit pixel count that wraps to 0, one that wraps to a small number, one just above and one exactly at 2^28 - 1
    def header_only(w, h, depth, ctype, interlace=0):
        return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, interlace))
                + chunk(b"IDAT", zlib.compress(bytes(64))) + chunk(b"IEND", b""))
    out["huge_wraps_to_zero_rgba16"] = header_only(0x80000000, 0x80000000, 16, 6)
    out["huge_65536x65536_rgb8"] = header_only(65536, 65536, 8, 2)
    out["huge_wraps_to_small_grey8"] = header_only(0x10001, 0x10000, 8, 0)       # 2^32 + 2^16 pixels: 65536 after the wrap
    out["huge_wraps_interlaced"] = header_only(0xFFFFFFFF, 0xFFFFFFFF, 8, 6, 1)
    out["big_above_limit_grey1"] = header_only(16384, 16384, 1, 0)               # 2^28 pixels
    out["big_at_limit_grey1"] = header_only(16383, 16385, 1, 0)                  # 2^28 - 1 pixels: sized, then the data is too short
    return out


NUMBERS = {
    # tokens are written verbatim into the document (no Python float formatting in between)
    "floats": ["0", "-0", "0.0", "-0.0", "1", "-1", "0.1", "-0.7", "0.2", "-3.9", "1e-45", "1.4e-45", "7e-46", "1e-38", "1.17549435e-38",
               "3e38", "3.4028234e38", "3.4028235e38", "-3.4028235e38", "16777216", "16777217", "16777218", "16777219", "-16777217",
               "16777217.0", "9007199254740993", "9007199254740993.0", "18446744073709551615", "18446744073709551616",
               "9223372036854775807", "9223372036854775808", "-9223372036854775808", "-9223372036854775809", "4294967297",
               "1E3", "1e+3", "1.5e-3", "123456789", "0.30000001192092896", "0.3", "2.2", "0.999999970197677612", "1.00000005960464478",
               "33554433", "1.0000000596046448", "5e-324", "1.7976931348623157e308", "12345.678e-2"],
    "ints": ["0", "-0", "0.7", "0.2", "-3.9", "-0.9", "0.999999", "1.0", "-1.0", "2.5", "-2.5", "1e2", "1.999e3", "3", "-7", "9.99", "-9.99",
             "2147483647", "-2147483648", "2147483647.5", "-2147483648.9", "4294967297", "-4294967297", "4294967296", "1e-30", "-1e-30",
             "123456.789", "16777217", "0.5", "-0.5", "12", "6", "-8", "9", "2", "-4"],
}


def main():
    if not os.path.exists(REF_CODEC):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
    os.makedirs(OUT, exist_ok=True)
    for f in os.listdir(OUT):
        os.remove(os.path.join(OUT, f))
    expected = {}
    tmp = os.path.join(OUT, "_tmp.bin")
    for name, data in sorted(cases().items()):
        path = os.path.join(OUT, name + ".png")
        open(path, "wb").write(data)
        subprocess.run([REF_CODEC, "png", path, tmp], check=True)
        raw = open(tmp, "rb").read()
        err, w, h = struct.unpack("<III", raw[:12])
        expected[f"png/{name}/err"] = np.uint32(err)
        expected[f"png/{name}/size"] = np.array([w, h], np.uint32)
        expected[f"png/{name}/rgba"] = np.frombuffer(raw[12:], np.uint8) if not err else np.zeros((0,), np.uint8)
    doc = '{"floats": [' + ", ".join(NUMBERS["floats"]) + '], "ints": [' + ", ".join(NUMBERS["ints"]) + "]}\n"
    jpath = os.path.join(OUT, "numbers.json")
    open(jpath, "w").write(doc)
    subprocess.run([REF_CODEC, "json", jpath, tmp], check=True)
    raw = open(tmp, "rb").read()
    n = struct.unpack("<I", raw[:4])[0]
    expected["json/floats"] = np.frombuffer(raw, np.float32, n, 4)
    m = struct.unpack("<I", raw[4 + 4 * n:8 + 4 * n])[0]
    expected["json/ints"] = np.frombuffer(raw, np.int32, m, 8 + 4 * n)
    expected["json/ints_as_float"] = np.frombuffer(raw, np.float32, m, 8 + 4 * n + 4 * m)
    assert n == len(NUMBERS["floats"]) and m == len(NUMBERS["ints"])
    os.remove(tmp)
    np.savez_compressed(os.path.join(OUT, "expected.npz"), **expected)
    errs = {k.split("/")[1]: int(v) for k, v in expected.items() if k.endswith("/err")}
    print(f"{len(errs)} PNG files, {sum(1 for v in errs.values() if v)} rejected by the reference: "
          + ", ".join(f"{k}={v}" for k, v in errs.items() if v))
    print("floats:", expected["json/floats"][:8], "... ints:", expected["json/ints"][:12])
    # the reference's framebuffer class through its fixed script of calls (oracle/ref_bitmap_harness.cpp)
    subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_bitmap"), os.path.join(HERE, "ref_bitmap.bin"),
                    os.path.join(HERE, "ref_config.json")], check=True)


if __name__ == "__main__":
    sys.exit(main())
