"""PNG -> (H, W, 3) uint8 texels for base-colour textures, in pure Python (zlib + numpy).

The reference sees glTF images through the gltf and image crates and `load_image` (src/scene.rs:87-114):
8-bit grey, grey+alpha, RGB or RGBA; `Image::value_at` (src/texture.rs:115-136) reads channels 0..2 of
`get_pixel`, i.e. (l, l, l) for grey images.  Palette images are expanded to RGB and 1/2/4-bit grey is scaled
to 0..255 by the decoder; 16-bit samples are refused, as the reference's loader panics on them.  This is the
decoder behind `load_from_gltf`; the C++ reader in the library has its own (csrc/png_decode.cpp) and the tests
compare the two.
"""
from __future__ import annotations

import struct
import zlib

import numpy as np

_ADAM7 = ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2))


def _unfilter(raw: bytes, pos: int, height: int, row_bytes: int, bpp: int) -> np.ndarray:
    rows = np.zeros((height, row_bytes), dtype=np.uint8)
    prev = np.zeros(row_bytes, dtype=np.int32)
    for y in range(height):
        start = pos + y * (row_bytes + 1)
        ftype = raw[start]
        line = np.frombuffer(raw, dtype=np.uint8, count=row_bytes, offset=start + 1).astype(np.int32)
        if ftype == 0:
            cur = line
        elif ftype == 2:
            cur = (line + prev) & 0xFF
        elif ftype in (1, 3, 4):
            cur = np.zeros(row_bytes, dtype=np.int32)
            for i in range(row_bytes):
                a = cur[i - bpp] if i >= bpp else 0
                b = prev[i]
                c = prev[i - bpp] if i >= bpp else 0
                if ftype == 1:
                    pred = a
                elif ftype == 3:
                    pred = (a + b) >> 1
                else:
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[i] = (line[i] + pred) & 0xFF
        else:
            raise ValueError("PNG: unknown filter type")
        rows[y] = cur
        prev = cur
    return rows


def decode_png_rgb8(data: bytes) -> np.ndarray:
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG file")
    off, idat, palette, header = 8, [], None, None
    while off + 12 <= len(data):
        (length,), ctype = struct.unpack(">I", data[off:off + 4]), data[off + 4:off + 8]
        body = data[off + 8:off + 8 + length]
        if ctype == b"IHDR":
            header = struct.unpack(">IIBBBBB", body[:13])
        elif ctype == b"PLTE":
            palette = np.frombuffer(body, dtype=np.uint8).reshape(-1, 3)
        elif ctype == b"IDAT":
            idat.append(body)
        elif ctype == b"IEND":
            break
        off += 12 + length
    if header is None:
        raise ValueError("PNG: missing header")
    w, h, depth, ctype, comp, filt, interlace = header
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}.get(ctype)
    if channels is None or comp or filt or interlace > 1:
        raise ValueError("PNG: unknown colour type or method")
    if depth == 16:
        raise ValueError("unsupported image format: 16-bit samples")
    if depth != 8 and not (ctype in (0, 3) and depth in (1, 2, 4)):
        raise ValueError("PNG: bit depth not allowed for the colour type")
    raw = zlib.decompress(b"".join(idat))
    bits_pp = channels * depth
    bpp = (bits_pp + 7) // 8
    out = np.zeros((h, w, 3), dtype=np.uint8)
    pos = 0
    for x0, y0, dx, dy in (_ADAM7 if interlace else ((0, 0, 1, 1),)):
        pw = (w - x0 + dx - 1) // dx if w > x0 else 0
        ph = (h - y0 + dy - 1) // dy if h > y0 else 0
        if not pw or not ph:
            continue
        row_bytes = (pw * bits_pp + 7) // 8
        rows = _unfilter(raw, pos, ph, row_bytes, bpp)
        pos += ph * (row_bytes + 1)
        if depth == 8:
            samples = rows.reshape(ph, pw, channels)
        else:
            bits = np.unpackbits(rows, axis=1)[:, :pw * depth].reshape(ph, pw, depth)
            samples = (bits * (1 << np.arange(depth - 1, -1, -1))).sum(axis=2, dtype=np.int64).astype(np.uint8)[..., None]
        if ctype == 3:
            if palette is None or samples.max() >= len(palette):
                raise ValueError("PNG: palette index out of range")
            px = palette[samples[..., 0]]
        elif ctype == 0:
            scale = {1: 255, 2: 85, 4: 17, 8: 1}[depth]
            px = np.repeat((samples[..., :1].astype(np.int32) * scale).astype(np.uint8), 3, axis=2)
        elif ctype == 4:
            px = np.repeat(samples[..., :1], 3, axis=2)
        else:
            px = samples[..., :3]
        out[y0::dy, x0::dx] = px
    if pos != len(raw):
        raise ValueError("PNG: image data does not inflate to the size the header implies")
    return out
