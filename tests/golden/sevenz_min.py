"""Minimal reader for the one 7z layout `/root/reference/gowalla_nevda.7z` uses (stdlib only: struct + lzma): an
LZMA-encoded header, single-coder LZMA / LZMA2 folders, solid sub-streams.  Fixture-generation tooling (no 7-Zip
tool or py7zr exists in the image); runs only in the build container, never on the GPU box."""
import struct, lzma, io, sys

def rnum(b):
    first = b.read(1)[0]
    mask = 0x80
    val = 0
    for i in range(8):
        if first & mask == 0:
            return val | ((first & (mask - 1)) << (8 * i))
        val |= b.read(1)[0] << (8 * i)
        mask >>= 1
    return val

def parse_streams(b):
    info = {}
    t = b.read(1)[0]
    if t == 0x06:
        info['packpos'] = rnum(b); n = rnum(b); info['npack'] = n
        t = b.read(1)[0]
        while t != 0:
            if t == 0x09:
                info['packsizes'] = [rnum(b) for _ in range(n)]
            elif t == 0x0a:
                alldef = b.read(1)[0]
                assert alldef
                b.read(4 * n)
            t = b.read(1)[0]
        t = b.read(1)[0]
    if t == 0x07:
        assert b.read(1)[0] == 0x0b
        nf = rnum(b); assert b.read(1)[0] == 0
        folders = []
        for _ in range(nf):
            nc = rnum(b); coders = []; nout_total = 0
            for _ in range(nc):
                fl = b.read(1)[0]
                cid = b.read(fl & 0xf)
                nin = nout = 1
                if fl & 0x10:
                    nin = rnum(b); nout = rnum(b)
                props = b''
                if fl & 0x20:
                    props = b.read(rnum(b))
                coders.append((cid, nin, nout, props)); nout_total += nout
            nbind = nout_total - 1
            binds = [(rnum(b), rnum(b)) for _ in range(nbind)]
            npacked = sum(c[1] for c in coders) - nbind
            if npacked > 1:
                [rnum(b) for _ in range(npacked)]
            folders.append(dict(coders=coders, binds=binds, nout=nout_total))
        info['folders'] = folders
        t = b.read(1)[0]
        while t != 0:
            if t == 0x0c:
                for f in folders:
                    f['unpack'] = [rnum(b) for _ in range(f['nout'])]
            elif t == 0x0a:
                alldef = b.read(1)[0]; assert alldef
                b.read(4 * nf)
            t = b.read(1)[0]
        t = b.read(1)[0]
    if t == 0x08:
        t = b.read(1)[0]
        nf = len(info['folders'])
        nstreams = [1] * nf
        sizes = None
        while t != 0:
            if t == 0x0d:
                nstreams = [rnum(b) for _ in range(nf)]
            elif t == 0x09:
                sizes = []
                for fi, ns in enumerate(nstreams):
                    s = [rnum(b) for _ in range(ns - 1)]
                    s.append(info['folders'][fi]['unpack'][-1] - sum(s))
                    sizes.append(s)
            elif t == 0x0a:
                alldef = b.read(1)[0]
                ndig = sum(nstreams)  # approx (streams w/o known crc)
                if not alldef:
                    raise NotImplementedError
                b.read(4 * ndig)
            t = b.read(1)[0]
        if sizes is None:
            sizes = [[f['unpack'][-1]] for f in info['folders']]
        info['substreams'] = sizes
        t = b.read(1)[0]
    assert t == 0, t
    return info

def decode_folder(f, raw, folder, start, packsize):
    cid, _, _, props = folder['coders'][0]
    assert len(folder['coders']) == 1, folder['coders']
    data = raw[start:start + packsize]
    if cid == b'\x03\x01\x01':
        lc = props[0] % 9; r = props[0] // 9; lp = r % 5; pb = r // 5
        ds = struct.unpack('<I', props[1:5])[0]
        d = lzma.LZMADecompressor(lzma.FORMAT_RAW, filters=[dict(id=lzma.FILTER_LZMA1, lc=lc, lp=lp, pb=pb, dict_size=ds)])
        return d.decompress(data, folder['unpack'][-1])
    if cid == b'\x21':
        bits = props[0]
        ds = 0xFFFFFFFF if bits == 40 else ((2 | (bits & 1)) << (bits // 2 + 11))
        d = lzma.LZMADecompressor(lzma.FORMAT_RAW, filters=[dict(id=lzma.FILTER_LZMA2, dict_size=ds)])
        return d.decompress(data, folder['unpack'][-1])
    raise NotImplementedError(cid)

def read_archive(path):
    raw = open(path, 'rb').read()
    assert raw[:6] == b"7z\xbc\xaf'\x1c"
    off, size, _ = struct.unpack('<QQI', raw[12:32])
    hdr = raw[32 + off:32 + off + size]
    b = io.BytesIO(hdr)
    t = b.read(1)[0]
    if t == 0x17:
        info = parse_streams(b)
        hdr = decode_folder(None, raw, info['folders'][0], 32 + info['packpos'], info['packsizes'][0])
        b = io.BytesIO(hdr)
        t = b.read(1)[0]
    assert t == 0x01
    t = b.read(1)[0]
    assert t == 0x04
    info = parse_streams(b)
    t = b.read(1)[0]
    assert t == 0x05
    nfiles = rnum(b)
    names = []; empty = [False] * nfiles
    while True:
        t = b.read(1)[0]
        if t == 0:
            break
        sz = rnum(b)
        blob = b.read(sz)
        if t == 0x11:
            assert blob[0] == 0
            s = blob[1:].decode('utf-16-le')
            names = s.split('\0')[:-1]
        elif t == 0x0e:
            bits = blob
            empty = [bool(bits[i // 8] & (0x80 >> (i % 8))) for i in range(nfiles)]
    pos = 32 + info['packpos']
    out = {}
    files = [n for n, e in zip(names, empty) if not e]
    fi = 0
    for k, folder in enumerate(info['folders']):
        data = decode_folder(None, raw, folder, pos, info['packsizes'][k])
        pos += info['packsizes'][k]
        o = 0
        for s in info['substreams'][k]:
            out[files[fi]] = data[o:o + s]; o += s; fi += 1
    return out

if __name__ == '__main__':
    files = read_archive(sys.argv[1])
    for n, d in files.items():
        print(n, len(d))
