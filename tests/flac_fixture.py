"""Test infrastructure: a small FLAC ENCODER written from the format specification (RFC 9639), so that the decoder behind the C ABI
(ssr_eval_amd/csrc/ssr_flac.h) can be exercised offline - the image has no FLAC tool, library or file.  It produces valid streams
that cycle through everything the decoder implements: CONSTANT / VERBATIM / FIXED (orders 0-4) / LPC (orders 1-32, quantised
coefficients) subframes, Rice residuals with 4- and 5-bit parameters, partition orders > 0, escaped (raw) partitions, wasted bits,
independent / left-side / right-side / mid-side stereo, 8 / 12 / 16 / 20 / 24-bit samples, every way of coding the block size, multi-byte
frame numbers, the MD5 signature, CRC-8 and CRC-16.  Encoding is exact by construction (residual = sample - integer prediction), so a
correct decoder returns the input bit for bit."""
import hashlib

import numpy as np


class BitWriter:
    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, value, bits):
        if bits == 0:
            return
        self.acc = (self.acc << bits) | (int(value) & ((1 << bits) - 1))
        self.n += bits
        while self.n >= 8:
            self.n -= 8
            self.out.append((self.acc >> self.n) & 0xff)
        self.acc &= (1 << self.n) - 1

    def put_signed(self, value, bits):
        self.put(int(value) & ((1 << bits) - 1), bits)

    def unary(self, q):
        while q >= 32:
            self.put(0, 32)
            q -= 32
        self.put(1, q + 1)

    def align(self):
        if self.n:
            self.put(0, 8 - self.n)

    def bytes(self):
        assert self.n == 0
        return bytes(self.out)


def crc8(data):
    c = 0
    for b in data:
        c ^= b
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xff if c & 0x80 else (c << 1) & 0xff
    return c


def crc16(data):
    c = 0
    for b in data:
        c ^= b << 8
        for _ in range(8):
            c = ((c << 1) ^ 0x8005) & 0xffff if c & 0x8000 else (c << 1) & 0xffff
    return c


def _utf8_number(v):
    if v < 0x80:
        return bytes([v])
    out, n = [], 0
    while True:
        n += 1
        lead_bits = 6 - n
        if v < (1 << (6 * n + lead_bits)):
            break
    for i in range(n):
        out.append(0x80 | ((v >> (6 * i)) & 0x3f))
    lead = ((0xff << (7 - n)) & 0xff) | (v >> (6 * n))
    return bytes([lead] + out[::-1])


def _rice_bits(res, k):
    u = np.where(res >= 0, 2 * res, -2 * res - 1)
    return int((u >> k).sum() + len(res) * (k + 1))


def _write_residual(w, res, order, bs, method, part_order, escape_part):
    """res: residuals for samples order .. bs-1."""
    pbits, esc = (4, 15) if method == 0 else (5, 31)
    w.put(method, 2)
    w.put(part_order, 4)
    parts = 1 << part_order
    i = 0
    for p in range(parts):
        count = (bs >> part_order) - (order if p == 0 else 0)
        r = res[i:i + count]
        i += count
        if p == escape_part and count > 0:
            raw = max(int(np.abs(r).max()).bit_length() + 1, 1) if len(r) else 1
            w.put(esc, pbits)
            w.put(raw, 5)
            for v in r:
                w.put_signed(int(v), raw)
            continue
        kmax = esc - 1
        k = min(range(0, kmax + 1), key=lambda kk: _rice_bits(r, kk)) if len(r) else 0
        w.put(k, pbits)
        for v in r:
            v = int(v)
            u = 2 * v if v >= 0 else -2 * v - 1
            w.unary(u >> k)
            w.put(u & ((1 << k) - 1), k)
    assert i == len(res)


FIXED = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}


def _predict(x, coefs, shift):
    """Integer prediction of x[t] from x[t-1-j] (python ints: no overflow), arithmetic shift."""
    order = len(coefs)
    pred = np.zeros(len(x), dtype=object)
    xs = [int(v) for v in x]
    for t in range(order, len(x)):
        acc = 0
        for j in range(order):
            acc += coefs[j] * xs[t - 1 - j]
        pred[t] = acc >> shift
    return pred


def _write_subframe(w, x, bps, kind, rng, opts):
    """x: int array [bs] (python-int safe), kind: 'constant' | 'verbatim' | ('fixed', o) | ('lpc', o)."""
    bs = len(x)
    wasted = 0
    if opts.get("wasted") and np.any(x != 0):
        wasted = opts["wasted"]
        assert np.all(x % (1 << wasted) == 0)
        x = x >> wasted
        bps -= wasted
    w.put(0, 1)
    if kind == "constant":
        w.put(0, 6)
    elif kind == "verbatim":
        w.put(1, 6)
    elif kind[0] == "fixed":
        w.put(8 + kind[1], 6)
    else:
        w.put(31 + kind[1], 6)
    if wasted:
        w.put(1, 1)
        w.unary(wasted - 1)
    else:
        w.put(0, 1)
    if kind == "constant":
        assert np.all(x == x[0])
        w.put_signed(int(x[0]), bps)
        return
    if kind == "verbatim":
        for v in x:
            w.put_signed(int(v), bps)
        return
    order = kind[1]
    for v in x[:order]:
        w.put_signed(int(v), bps)
    if kind[0] == "fixed":
        coefs, shift = FIXED[order], 0
    else:
        prec = opts.get("lpc_precision", 12)
        shift = opts.get("lpc_shift", 9)
        xf = np.asarray(x, dtype=np.float64)
        # a least-squares predictor, quantised (any integer coefficients give a valid stream; good ones keep the residual small)
        if bs > 2 * order + 2 and np.any(xf != 0):
            A = np.stack([xf[order - 1 - j:bs - 1 - j] for j in range(order)], axis=1)
            sol = np.linalg.lstsq(A, xf[order:], rcond=None)[0]
        else:
            sol = np.zeros(order)
        lim = (1 << (prec - 1)) - 1
        coefs = [int(np.clip(np.rint(c * (1 << shift)), -lim - 1, lim)) for c in sol]
        w.put(prec - 1, 4)
        w.put_signed(shift, 5)
        for c in coefs:
            w.put_signed(c, prec)
    pred = _predict(x, coefs, shift)
    res = np.array([int(x[t]) - int(pred[t]) for t in range(order, bs)], dtype=object)
    res64 = np.array([int(v) for v in res], dtype=np.int64)
    po = opts.get("partition_order", 0)
    while po > 0 and ((bs >> po) << po != bs or (bs >> po) < order):
        po -= 1
    _write_residual(w, res64, order, bs, opts.get("rice_method", 0), po, opts.get("escape_partition", -1))


def encode(x, sample_rate, bits=16, block_size=1152, plan=None, md5=True, seed=0, variable=False):
    """x: int array [n, channels] -> bytes of a FLAC stream.  plan(frame_index, n_channels) -> dict with optional keys
    'stereo' (0 independent, 8 left-side, 9 right-side, 10 mid-side), 'kinds' (one per channel) and the subframe options of
    _write_subframe; default: a rotation through everything."""
    x = np.asarray(x)
    if x.ndim == 1:
        x = x[:, None]
    n, nch = x.shape
    rng = np.random.default_rng(seed)
    frames = bytearray()
    kinds_cycle = ["verbatim", ("fixed", 0), ("fixed", 1), ("fixed", 2), ("fixed", 3), ("fixed", 4), ("lpc", 1), ("lpc", 2), ("lpc", 8),
                   ("lpc", 12), ("lpc", 32)]
    pos, fi = 0, 0
    min_fs, max_fs = 1 << 30, 0
    while pos < n:
        bs = min(block_size, n - pos)
        blk = x[pos:pos + bs].astype(np.int64)
        p = plan(fi, nch) if plan else {}
        stereo = p.get("stereo", [0, 8, 9, 10][fi % 4] if nch == 2 else 0)
        w = BitWriter()
        w.put(0x3ffe, 14)
        w.put(0, 1)
        w.put(1 if variable else 0, 1)
        std = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12, 8192: 13, 16384: 14, 32768: 15}
        if bs in std and not p.get("explicit_block_size"):
            bcode = std[bs]
        else:
            bcode = 6 if bs <= 256 else 7
        w.put(bcode, 4)
        src = {88200: 1, 176400: 2, 192000: 3, 8000: 4, 16000: 5, 22050: 6, 24000: 7, 32000: 8, 44100: 9, 48000: 10, 96000: 11}
        scode = src.get(sample_rate, 0) if not p.get("explicit_sample_rate") else (13 if sample_rate < 65536 else 0)
        w.put(scode, 4)
        w.put(stereo if nch == 2 and stereo >= 8 else nch - 1, 4)
        w.put({8: 1, 12: 2, 16: 4, 20: 5, 24: 6, 32: 7}.get(bits, 0) if p.get("explicit_bits", True) else 0, 3)
        w.put(0, 1)
        for b in _utf8_number(pos if variable else fi):
            w.put(b, 8)
        if bcode == 6:
            w.put(bs - 1, 8)
        elif bcode == 7:
            w.put(bs - 1, 16)
        if scode == 13:
            w.put(sample_rate, 16)
        w.put(crc8(bytes(w.out)), 8)
        chans = [blk[:, c] for c in range(nch)]
        sub_bits = [bits] * nch
        if nch == 2 and stereo == 8:
            chans = [blk[:, 0], blk[:, 0] - blk[:, 1]]
            sub_bits = [bits, bits + 1]
        elif nch == 2 and stereo == 9:
            chans = [blk[:, 0] - blk[:, 1], blk[:, 1]]
            sub_bits = [bits + 1, bits]
        elif nch == 2 and stereo == 10:
            chans = [(blk[:, 0] + blk[:, 1]) >> 1, blk[:, 0] - blk[:, 1]]
            sub_bits = [bits, bits + 1]
        kinds = p.get("kinds")
        for c in range(nch):
            ch = chans[c]
            if kinds is not None:
                kind = kinds[c]
            elif np.all(ch == ch[0]):
                kind = "constant"
            else:
                kind = kinds_cycle[(fi + 3 * c) % len(kinds_cycle)]
            if kind != "constant" and kind != "verbatim" and kind[1] > bs:
                kind = "verbatim"
            opts = dict(p)
            opts.setdefault("rice_method", (fi // 2) % 2)
            opts.setdefault("partition_order", fi % 4)
            opts.setdefault("escape_partition", 0 if fi % 7 == 3 else -1)
            _write_subframe(w, ch, sub_bits[c], kind, rng, opts)
        w.align()
        body = bytes(w.out)
        c16 = crc16(body)
        frame = body + bytes([c16 >> 8, c16 & 0xff])
        min_fs, max_fs = min(min_fs, len(frame)), max(max_fs, len(frame))
        frames += frame
        pos += bs
        fi += 1
    # STREAMINFO
    si = BitWriter()
    si.put(block_size if n >= block_size else max(n, 16), 16)
    si.put(block_size if n >= block_size else max(n, 16), 16)
    si.put(min_fs if n else 0, 24)
    si.put(max_fs, 24)
    si.put(sample_rate, 20)
    si.put(nch - 1, 3)
    si.put(bits - 1, 5)
    si.put(n, 36)
    bytes_per = (bits + 7) // 8
    if md5:
        raw = bytearray()
        flat = x.reshape(-1)
        for v in flat:
            raw += (int(v) & ((1 << (8 * bytes_per)) - 1)).to_bytes(bytes_per, "little")
        digest = hashlib.md5(bytes(raw)).digest()
    else:
        digest = bytes(16)
    for b in digest:
        si.put(b, 8)
    info = si.bytes()
    assert len(info) == 34
    pad = bytes([0x81, 0, 0, 8]) + bytes(8)                       # a PADDING block (type 1), last
    return b"fLaC" + bytes([0x00, 0, 0, 34]) + info + pad + bytes(frames)
