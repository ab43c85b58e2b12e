"""Baseline JPEG decode with the arithmetic of the reference's image decoder (its vendored stb_image, called from
Texture.cu:23 with 0 requested channels).  TEST INFRASTRUCTURE ONLY (oracle side of the loader).

A second, independent restatement (numpy; the product's is C++ in dustraytracer_amd/csrc/jpeg_decode.cpp) of the
choices a JPEG decoder is free to make and stb_image makes this way:
  * inverse DCT: the integer LLM algorithm with 12-bit constants, column pass rounded to 10 bits, row pass to 17 with the
    +128 level shift folded in, dequantised coefficients kept in 16 bits;
  * chroma upsampling: 3:1 triangle filter horizontally / vertically / both, nearest for other factors, with stb's choice
    of "near" and "far" rows;
  * YCbCr -> RGB in 20-bit fixed point, the green Cb term truncated to its upper 16 bits.
Both restatements are pinned by tests/golden/jpeg_ref.json, which the reference's own compiled decoder produced.
"""
import numpy as np

_DEZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21,
                      28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61,
                      54, 47, 55, 62, 63] + [63] * 15)


class JpegError(ValueError):
    pass


def looks_like_jpeg(data):
    return len(data) >= 3 and data[0] == 0xFF and data[1] == 0xD8 and data[2] == 0xFF


def _huffman_lut(counts, values):
    """peek16 -> (code length, symbol) tables."""
    length = np.zeros(65536, np.uint8)
    symbol = np.zeros(65536, np.uint8)
    code, k = 0, 0
    for ln in range(1, 17):
        for _ in range(counts[ln - 1]):
            lo = code << (16 - ln)
            length[lo:lo + (1 << (16 - ln))] = ln
            symbol[lo:lo + (1 << (16 - ln))] = values[k]
            code += 1
            k += 1
        code <<= 1
    return length.tolist(), symbol.tolist()


def _f2f(x):
    return int(float(np.float32(x)) * 4096 + 0.5)          # (int)(float constant * 4096 + 0.5): truncation toward zero


def _idct_1d(s0, s1, s2, s3, s4, s5, s6, s7):
    p2, p3 = s2, s6
    p1 = (p2 + p3) * _f2f(0.5411961)
    t2 = p1 + p3 * _f2f(-1.847759065)
    t3 = p1 + p2 * _f2f(0.765366865)
    p2, p3 = s0, s4
    t0 = (p2 + p3) * 4096
    t1 = (p2 - p3) * 4096
    x0, x3, x1, x2 = t0 + t3, t0 - t3, t1 + t2, t1 - t2
    t0, t1, t2, t3 = s7, s5, s3, s1
    p3, p4, p1, p2 = t0 + t2, t1 + t3, t0 + t3, t1 + t2
    p5 = (p3 + p4) * _f2f(1.175875602)
    t0 = t0 * _f2f(0.298631336)
    t1 = t1 * _f2f(2.053119869)
    t2 = t2 * _f2f(3.072711026)
    t3 = t3 * _f2f(1.501321110)
    p1 = p5 + p1 * _f2f(-0.899976223)
    p2 = p5 + p2 * _f2f(-2.562915447)
    p3 = p3 * _f2f(-1.961570560)
    p4 = p4 * _f2f(-0.390180644)
    return x0, x1, x2, x3, t0 + p1 + p3, t1 + p2 + p4, t2 + p2 + p3, t3 + p1 + p4


def _idct_blocks(coef):
    """coef: int16 [N, 64] (natural order, dequantised) -> uint8 [N, 8, 8]."""
    d = coef.astype(np.int64).reshape(-1, 8, 8)                 # [n, row, col]
    x0, x1, x2, x3, t0, t1, t2, t3 = _idct_1d(*[d[:, r, :] for r in range(8)])       # columns: along rows r
    x0, x1, x2, x3 = x0 + 512, x1 + 512, x2 + 512, x3 + 512
    v = np.stack([(x0 + t3) >> 10, (x1 + t2) >> 10, (x2 + t1) >> 10, (x3 + t0) >> 10,
                  (x3 - t0) >> 10, (x2 - t1) >> 10, (x1 - t2) >> 10, (x0 - t3) >> 10], axis=1)     # [n, row, col]
    x0, x1, x2, x3, t0, t1, t2, t3 = _idct_1d(*[v[:, :, c] for c in range(8)])       # rows: along columns c
    bias = 65536 + (128 << 17)
    x0, x1, x2, x3 = x0 + bias, x1 + bias, x2 + bias, x3 + bias
    o = np.stack([(x0 + t3) >> 17, (x1 + t2) >> 17, (x2 + t1) >> 17, (x3 + t0) >> 17,
                  (x3 - t0) >> 17, (x2 - t1) >> 17, (x1 - t2) >> 17, (x0 - t3) >> 17], axis=2)     # [n, row, col]
    return np.clip(o, 0, 255).astype(np.uint8)


def _near_far_rows(height, vs, comp_rows):
    """stb's per-output-row choice of the two source rows of a component (load_jpeg_image's ystep / ypos walk)."""
    near, far = [], []
    ystep, ypos, line0, line1 = vs >> 1, 0, 0, 0
    for _ in range(height):
        y_bot = ystep >= (vs >> 1)
        near.append(line1 if y_bot else line0)
        far.append(line0 if y_bot else line1)
        ystep += 1
        if ystep >= vs:
            ystep = 0
            line0 = line1
            ypos += 1
            if ypos < comp_rows:
                line1 += 1
    return np.array(near), np.array(far)


def _upsample(plane, comp_w, comp_h, hs, vs, width, height):
    w = (width + hs - 1) // hs
    near_i, far_i = _near_far_rows(height, vs, comp_h)
    near = plane[near_i, :w].astype(np.int32)
    far = plane[far_i, :w].astype(np.int32)
    if hs == 1 and vs == 1:
        out = near
    elif hs == 1 and vs == 2:
        out = (3 * near + far + 2) >> 2
    elif hs == 2 and vs == 1:
        out = np.zeros((height, 2 * w), np.int32)
        if w == 1:
            out[:, 0] = out[:, 1] = near[:, 0]
        else:
            out[:, 0] = near[:, 0]
            out[:, 1] = (near[:, 0] * 3 + near[:, 1] + 2) >> 2
            n = 3 * near[:, 1:w - 1] + 2
            out[:, 2:2 * w - 2:2] = (n + near[:, 0:w - 2]) >> 2
            out[:, 3:2 * w - 1:2] = (n + near[:, 2:w]) >> 2
            out[:, 2 * w - 2] = (near[:, w - 2] * 3 + near[:, w - 1] + 2) >> 2
            out[:, 2 * w - 1] = near[:, w - 1]
    elif hs == 2 and vs == 2:
        t = 3 * near + far
        out = np.zeros((height, 2 * w), np.int32)
        if w == 1:
            out[:, 0] = out[:, 1] = (t[:, 0] + 2) >> 2
        else:
            out[:, 0] = (t[:, 0] + 2) >> 2
            out[:, 1:2 * w - 1:2] = (3 * t[:, :-1] + t[:, 1:] + 8) >> 4
            out[:, 2:2 * w:2] = (3 * t[:, 1:] + t[:, :-1] + 8) >> 4
            out[:, 2 * w - 1] = (t[:, w - 1] + 2) >> 2
    else:
        out = np.repeat(near, hs, axis=1)
    return out[:, :width].astype(np.uint8)


def _fixed20(x):
    return int(np.float32(x) * np.float32(4096.0) + np.float32(0.5)) << 8


def _ycbcr_to_rgb(y, cb, cr):
    y_fixed = (y.astype(np.int64) << 20) + (1 << 19)
    cr = cr.astype(np.int64) - 128
    cb = cb.astype(np.int64) - 128
    r = y_fixed + cr * _fixed20(1.40200)
    g = y_fixed + cr * -_fixed20(0.71414) + (((cb * -_fixed20(0.34414)) & 0xFFFFFFFF) & 0xFFFF0000)
    g = ((g & 0xFFFFFFFF) ^ 0x80000000) - 0x80000000                      # wrap to int32 as the C expression does
    b = y_fixed + cb * _fixed20(1.77200)
    return np.stack([np.clip(c >> 20, 0, 255) for c in (r, g, b)], axis=-1).astype(np.uint8)


def decode(data):
    """-> uint8 [H, W, C] with C = 3 (colour) or 1 (greyscale), as stbi_load_from_memory(..., 0) returns it."""
    data = bytes(data)
    if not looks_like_jpeg(data):
        raise JpegError("not a JPEG stream")
    be16 = lambda i: (data[i] << 8) | data[i + 1]
    dequant, dc_tab, ac_tab = {}, {}, {}
    comps, restart_interval = [], 0
    jfif, adobe_transform, rgb_ids = False, -1, 0
    width = height = h_max = v_max = mcu_x = mcu_y = 0
    coef = None
    p, n = 2, len(data)
    while True:
        while p < n and data[p] != 0xFF:
            p += 1
        while p < n and data[p] == 0xFF:
            p += 1
        if p >= n:
            raise JpegError("truncated (no EOI)")
        m = data[p]
        p += 1
        if m == 0xD9:
            break
        if m == 0x01 or 0xD0 <= m <= 0xD7:
            continue
        ln = be16(p)
        s, s_end = p + 2, p + ln
        p = s_end
        if m == 0xDB:
            while s < s_end:
                pq, tq = data[s] >> 4, data[s] & 15
                s += 1
                q = np.zeros(64, np.int64)
                for i in range(64):
                    q[_DEZIGZAG[i]] = be16(s) if pq else data[s]
                    s += 2 if pq else 1
                dequant[tq] = q
        elif m == 0xC4:
            while s < s_end:
                tc, th = data[s] >> 4, data[s] & 15
                counts = list(data[s + 1:s + 17])
                nv = sum(counts)
                (ac_tab if tc else dc_tab)[th] = _huffman_lut(counts, list(data[s + 17:s + 17 + nv]))
                s += 17 + nv
        elif m in (0xC0, 0xC1):
            if data[s] != 8:
                raise JpegError("only 8-bit samples are supported")
            height, width, nc = be16(s + 1), be16(s + 3), data[s + 5]
            if nc not in (1, 3):
                raise JpegError("only 1- or 3-component images are supported")
            for i in range(nc):
                cid, hv, tq = data[s + 6 + 3 * i], data[s + 7 + 3 * i], data[s + 8 + 3 * i]
                comps.append(dict(id=cid, h=hv >> 4, v=hv & 15, tq=tq, pred=0))
                if cid == b"RGB"[i]:
                    rgb_ids += 1
            h_max, v_max = max(c["h"] for c in comps), max(c["v"] for c in comps)
            mcu_x = (width + 8 * h_max - 1) // (8 * h_max)
            mcu_y = (height + 8 * v_max - 1) // (8 * v_max)
            for c in comps:
                c["x"] = (width * c["h"] + h_max - 1) // h_max
                c["y"] = (height * c["v"] + v_max - 1) // v_max
                c["bw"], c["bh"] = mcu_x * c["h"], mcu_y * c["v"]
                c["coef"] = np.zeros((c["bh"], c["bw"], 64), np.int64)
        elif m == 0xC2:
            raise JpegError("progressive JPEG is not supported")
        elif m in (0xC3, 0xC5, 0xC6, 0xC7, 0xC9, 0xCA, 0xCB, 0xCD, 0xCE, 0xCF):
            raise JpegError("lossless / hierarchical / arithmetic-coded JPEG is not supported")
        elif m == 0xDD:
            restart_interval = be16(s)
        elif m == 0xE0:
            jfif = jfif or data[s:s + 5] == b"JFIF\0"
        elif m == 0xEE:
            if data[s:s + 6] == b"Adobe\0" and s_end - s >= 12:
                adobe_transform = data[s + 11]
        elif m == 0xDA:
            ns = data[s]
            order = []
            for i in range(ns):
                c = next(c for c in comps if c["id"] == data[s + 1 + 2 * i])
                c["td"], c["ta"] = data[s + 2 + 2 * i] >> 4, data[s + 2 + 2 * i] & 15
                c["pred"] = 0
                order.append(c)
            p = _decode_scan(data, p, order, comps, dc_tab, ac_tab, restart_interval, mcu_x, mcu_y)
    if not comps:
        raise JpegError("no frame header")
    planes = []
    for c in comps:
        co = (c["coef"].reshape(-1, 64) * dequant[c["tq"]][None, :]).astype(np.int16)      # 16-bit storage of the products
        blocks = _idct_blocks(co).reshape(c["bh"], c["bw"], 8, 8)
        plane = blocks.transpose(0, 2, 1, 3).reshape(c["bh"] * 8, c["bw"] * 8)
        planes.append(_upsample(plane, c["x"], c["y"], h_max // c["h"], v_max // c["v"], width, height))
    if len(comps) == 1:
        return planes[0][:, :, None].copy()
    if rgb_ids == 3 or (adobe_transform == 0 and not jfif):
        return np.stack(planes, axis=-1)
    return _ycbcr_to_rgb(planes[0], planes[1], planes[2])


def _decode_scan(data, start, order, comps, dc_tab, ac_tab, restart_interval, mcu_x, mcu_y):
    """Huffman-decodes one scan into the components' coefficient arrays (quantised, natural order). Returns the position
    after the entropy-coded data."""
    # entropy-coded bytes: un-stuff FF00, stop at any other marker; restart markers split the data into intervals
    n = len(data)
    intervals, cur, p = [], bytearray(), start
    end_pos = n
    while p < n:
        b = data[p]
        if b != 0xFF:
            cur.append(b)
            p += 1
            continue
        q = p + 1
        while q < n and data[q] == 0xFF:
            q += 1
        mk = data[q] if q < n else 0xD9
        if mk == 0:
            cur.append(0xFF)
            p = q + 1
        elif 0xD0 <= mk <= 0xD7:
            intervals.append(bytes(cur))
            cur = bytearray()
            p = q + 1
        else:
            end_pos = p
            break
    intervals.append(bytes(cur))

    if len(order) == 1:                                   # decode order of (component, block x, block y) within one MCU
        c = order[0]
        w, h = (c["x"] + 7) >> 3, (c["y"] + 7) >> 3
        mcus = [(i, j) for j in range(h) for i in range(w)]
        per_mcu = lambda i, j: [(c, i, j)]
    else:
        mcus = [(i, j) for j in range(mcu_y) for i in range(mcu_x)]
        per_mcu = lambda i, j: [(c, i * c["h"] + x, j * c["v"] + y) for c in order for y in range(c["v"]) for x in range(c["h"])]
    dezig = _DEZIGZAG.tolist()
    todo_reset = restart_interval if restart_interval else 0x7FFFFFFF
    it = iter(intervals)
    mcu_index = 0
    for seg in it:
        buf = seg + b"\0" * 8                               # zero bits after the data, as stb feeds them
        words = [(buf[i] << 24) | (buf[i + 1] << 16) | (buf[i + 2] << 8) | buf[i + 3] for i in range(len(buf) - 3)]
        pos = 0
        for c in comps:
            c["pred"] = 0
        todo = todo_reset
        while mcu_index < len(mcus) and todo > 0:
            i, j = mcus[mcu_index]
            for c, bx, by in per_mcu(i, j):
                dlen, dsym = dc_tab[c["td"]]
                alen, asym = ac_tab[c["ta"]]
                out = c["coef"][by, bx]
                k16 = (words[pos >> 3] >> (16 - (pos & 7))) & 0xFFFF
                ln = dlen[k16]
                if ln == 0:
                    raise JpegError("bad Huffman code")
                t = dsym[k16]
                pos += ln
                if t:
                    v = (words[pos >> 3] >> (32 - t - (pos & 7))) & ((1 << t) - 1)
                    pos += t
                    if v < (1 << (t - 1)):
                        v += 1 - (1 << t)
                    c["pred"] += v
                out[0] = c["pred"]
                k = 1
                while k < 64:
                    k16 = (words[pos >> 3] >> (16 - (pos & 7))) & 0xFFFF
                    ln = alen[k16]
                    if ln == 0:
                        raise JpegError("bad Huffman code")
                    rs = asym[k16]
                    pos += ln
                    ssss, run = rs & 15, rs >> 4
                    if ssss == 0:
                        if rs != 0xF0:
                            break
                        k += 16
                    else:
                        k += run
                        v = (words[pos >> 3] >> (32 - ssss - (pos & 7))) & ((1 << ssss) - 1)
                        pos += ssss
                        if v < (1 << (ssss - 1)):
                            v += 1 - (1 << ssss)
                        out[dezig[k]] = v
                        k += 1
            mcu_index += 1
            todo -= 1
        if mcu_index >= len(mcus):
            break
    return end_pos
