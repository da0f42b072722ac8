"""Reader (and a small writer) for TensorFlow's tensor-bundle checkpoint format, host side only.

The reference saves its stylizer with ``tf.train.Saver().save(sess, 'saved_<name>', global_step=0)``
(full_graph_stylizer_network.py:183, :231) and restores it with ``saver.restore`` (:366-367); the
three models its Django view names (views.py:23-29) exist only in that format:

    saved_<name>-0.index                  an SSTable (the LevelDB table format) key -> protobuf
    saved_<name>-0.data-00000-of-00001    the raw tensor bytes

Nothing here needs TensorFlow.  The format as published (tensorflow/core/util/tensor_bundle,
tensorflow/core/lib/io/{table,block,format}, tensorflow/core/protobuf/tensor_bundle.proto):

  * table file = blocks + 48-byte footer (metaindex handle, index handle, padding, magic
    0xdb4775248b80fb57 little endian); a handle is (varint64 offset, varint64 size);
  * block = entries (varint32 shared, varint32 non_shared, varint32 value_len, key suffix, value)
    + uint32 restart offsets + uint32 restart count, followed by a 5-byte trailer: compression
    type (0 = none; the bundle writer never compresses) and the masked CRC32C of block + type;
  * key "" holds BundleHeaderProto (num_shards = 1, endianness = 2, version = 3); every other key
    is a variable name holding BundleEntryProto: dtype = 1, shape = 2 (TensorShapeProto: dim = 2
    {size = 1}), shard_id = 3, offset = 4, size = 5, crc32c = 6 (fixed32, masked CRC32C of the
    bytes), slices = 7 (partitioned variables; not used by the reference);
  * masked crc = rotate_right(crc, 15) + 0xa282ead8.

No checkpoint of the reference is available to test against (they are git-ignored there), so the
reader is pinned by the published constants (CRC32C check value, table magic) and by round trips
through the writer below, which follows the same description: parity unpinned.
"""
import os
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
_MASK_DELTA = 0xa282ead8
# DataType enum values of tensorflow/core/framework/types.proto that a Saver of float models emits
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64, 4: np.uint8, 6: np.int8, 5: np.int16, 10: np.bool_}
_DTYPE_CODES = {np.dtype(v): k for k, v in _DTYPES.items()}


class BundleError(ValueError):
    pass


# ------------------------------------------------------------------ CRC32C (Castagnoli)
def _make_table():
    poly, table = 0x82f63b78, []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ poly if c & 1 else c >> 1
        table.append(c)
    return table


_CRC_TABLE = _make_table()


def crc32c(data, crc=0):
    c = crc ^ 0xffffffff
    t = _CRC_TABLE
    for b in bytes(data):
        c = t[(c ^ b) & 0xff] ^ (c >> 8)
    return c ^ 0xffffffff


def mask_crc(crc):
    return (((crc >> 15) | (crc << 17)) + _MASK_DELTA) & 0xffffffff


def unmask_crc(masked):
    rot = (masked - _MASK_DELTA) & 0xffffffff
    return ((rot >> 17) | (rot << 15)) & 0xffffffff


# ------------------------------------------------------------------ varints and protobuf fields
def _get_varint(buf, pos):
    result = shift = 0
    while True:
        if pos >= len(buf):
            raise BundleError('truncated varint')
        b = buf[pos]
        pos += 1
        result |= (b & 0x7f) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 63:
            raise BundleError('varint too long')


def _put_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7f
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _proto_fields(buf):
    """Yields (field number, wire type, value) of one protobuf message; value is an int for
    varint / fixed fields and bytes for length-delimited ones."""
    pos = 0
    while pos < len(buf):
        tag, pos = _get_varint(buf, pos)
        num, wire = tag >> 3, tag & 7
        if wire == 0:
            val, pos = _get_varint(buf, pos)
        elif wire == 1:
            val = struct.unpack_from('<Q', buf, pos)[0]
            pos += 8
        elif wire == 2:
            ln, pos = _get_varint(buf, pos)
            val = bytes(buf[pos:pos + ln])
            if len(val) != ln:
                raise BundleError('truncated protobuf field')
            pos += ln
        elif wire == 5:
            val = struct.unpack_from('<I', buf, pos)[0]
            pos += 4
        else:
            raise BundleError('unsupported protobuf wire type %d' % wire)
        yield num, wire, val


def _signed64(v):
    return v - (1 << 64) if v >= 1 << 63 else v


def _parse_shape(buf):
    dims = []
    for num, _, val in _proto_fields(buf):
        if num == 2:      # Dim
            size = 0
            for n2, _, v2 in _proto_fields(val):
                if n2 == 1:
                    size = _signed64(v2)
            dims.append(size)
        elif num == 3 and val:
            raise BundleError('tensor of unknown rank in a checkpoint')
    return tuple(dims)


def _parse_entry(buf):
    e = {'dtype': 0, 'shape': (), 'shard_id': 0, 'offset': 0, 'size': 0, 'crc32c': None, 'slices': 0}
    for num, _, val in _proto_fields(buf):
        if num == 1:
            e['dtype'] = val
        elif num == 2:
            e['shape'] = _parse_shape(val)
        elif num == 3:
            e['shard_id'] = val
        elif num == 4:
            e['offset'] = val
        elif num == 5:
            e['size'] = val
        elif num == 6:
            e['crc32c'] = val
        elif num == 7:
            e['slices'] += 1
    return e


def _parse_header(buf):
    h = {'num_shards': 0, 'endianness': 0}
    for num, _, val in _proto_fields(buf):
        if num == 1:
            h['num_shards'] = val
        elif num == 2:
            h['endianness'] = val
    return h


# ------------------------------------------------------------------ table reader
def _read_block(data, offset, size, verify=True):
    end = offset + size
    if end + 5 > len(data):
        raise BundleError('block handle (%d, %d) runs past the end of the file' % (offset, size))
    ctype = data[end]
    if verify:
        stored = struct.unpack_from('<I', data, end + 1)[0]
        if unmask_crc(stored) != crc32c(data[offset:end + 1]):
            raise BundleError('block checksum mismatch at offset %d' % offset)
    if ctype != 0:
        raise BundleError('compressed table block (type %d): tensor-bundle indexes are written uncompressed' % ctype)
    return data[offset:end]


def _block_entries(block):
    if len(block) < 4:
        raise BundleError('table block too small')
    restarts = struct.unpack_from('<I', block, len(block) - 4)[0]
    limit = len(block) - 4 - 4 * restarts
    if limit < 0:
        raise BundleError('bad restart count %d' % restarts)
    pos, key = 0, b''
    while pos < limit:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        if shared > len(key) or pos + non_shared + vlen > limit:
            raise BundleError('corrupt table entry')
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        yield key, bytes(block[pos:pos + vlen])
        pos += vlen


def read_table(path, verify=True):
    """All (key bytes, value bytes) pairs of an SSTable file, in key order."""
    with open(path, 'rb') as f:
        data = f.read()
    if len(data) < 48:
        raise BundleError('%s is too short to be a table file' % path)
    footer = data[-48:]
    if struct.unpack('<Q', footer[40:])[0] != TABLE_MAGIC:
        raise BundleError('%s: bad table magic' % path)
    pos = 0
    _, pos = _get_varint(footer, pos)       # metaindex offset
    _, pos = _get_varint(footer, pos)       # metaindex size
    ioff, pos = _get_varint(footer, pos)
    isize, pos = _get_varint(footer, pos)
    out = []
    for _, handle in _block_entries(_read_block(data, ioff, isize, verify)):
        boff, p2 = _get_varint(handle, 0)
        bsize, _ = _get_varint(handle, p2)
        out.extend(_block_entries(_read_block(data, boff, bsize, verify)))
    return out


# ------------------------------------------------------------------ bundle reader
def read_bundle(prefix, names=None, verify_data=False):
    """{variable name: ndarray} of the checkpoint whose files start with `prefix` (the path
    without '.index').  names: restrict to these keys (missing ones raise).  verify_data checks
    the CRC32C of every tensor read (slow in pure Python for large tensors)."""
    entries = read_table(prefix + '.index')
    if not entries or entries[0][0] != b'':
        raise BundleError('%s.index has no bundle header entry' % prefix)
    header = _parse_header(entries[0][1])
    if header['endianness'] != 0:
        raise BundleError('big-endian bundle')
    table = {k.decode('utf-8'): _parse_entry(v) for k, v in entries[1:]}
    wanted = list(table) if names is None else list(names)
    out, shards = {}, {}
    for name in wanted:
        if name not in table:
            raise KeyError('%s: no variable %r in the checkpoint' % (prefix, name))
        e = table[name]
        if e['slices']:
            raise BundleError('%s is a partitioned variable (not produced by the reference)' % name)
        if e['dtype'] not in _DTYPES:
            raise BundleError('%s: unsupported dtype code %d' % (name, e['dtype']))
        dt = np.dtype(_DTYPES[e['dtype']])
        count = int(np.prod(e['shape'], dtype=np.int64)) if e['shape'] else 1
        if count * dt.itemsize != e['size']:
            raise BundleError('%s: shape %s does not match %d bytes' % (name, e['shape'], e['size']))
        sid = e['shard_id']
        if sid not in shards:
            shards[sid] = '%s.data-%05d-of-%05d' % (prefix, sid, max(header['num_shards'], 1))
        with open(shards[sid], 'rb') as f:
            f.seek(e['offset'])
            raw = f.read(e['size'])
        if len(raw) != e['size']:
            raise BundleError('%s: data file is truncated' % name)
        if verify_data and e['crc32c'] is not None and unmask_crc(e['crc32c']) != crc32c(raw):
            raise BundleError('%s: tensor checksum mismatch' % name)
        out[name] = np.frombuffer(raw, dtype=dt).reshape(e['shape']).copy()
    return out


def list_bundle(prefix):
    """{variable name: (dtype, shape)} without touching the data file."""
    entries = read_table(prefix + '.index')
    out = {}
    for k, v in entries[1:]:
        e = _parse_entry(v)
        out[k.decode('utf-8')] = (np.dtype(_DTYPES[e['dtype']]) if e['dtype'] in _DTYPES else None, e['shape'])
    return out


# ------------------------------------------------------------------ writer (tests, export)
def _field(num, wire, payload):
    return _put_varint((num << 3) | wire) + payload


def _encode_entry(arr, offset, crc):
    shape = b''.join(_field(2, 2, _put_varint(len(d)) + d)
                     for d in (_field(1, 0, _put_varint(int(s))) for s in arr.shape))
    msg = _field(1, 0, _put_varint(_DTYPE_CODES[arr.dtype]))
    msg += _field(2, 2, _put_varint(len(shape)) + shape)
    if offset:
        msg += _field(4, 0, _put_varint(offset))
    msg += _field(5, 0, _put_varint(arr.nbytes))
    msg += _field(6, 5, struct.pack('<I', crc))
    return msg


class _BlockBuilder:
    def __init__(self, restart_interval=16):
        self.buf, self.restarts, self.count, self.last = bytearray(), [0], 0, b''
        self.interval = restart_interval

    def add(self, key, value):
        shared = 0
        if self.count and self.count % self.interval == 0:
            self.restarts.append(len(self.buf))
        elif self.count:
            while shared < min(len(key), len(self.last)) and key[shared] == self.last[shared]:
                shared += 1
        self.buf += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value))
        self.buf += key[shared:] + value
        self.last, self.count = key, self.count + 1

    def finish(self):
        return bytes(self.buf) + b''.join(struct.pack('<I', r) for r in self.restarts) + struct.pack('<I', len(self.restarts))


def _emit_block(out, contents):
    offset = len(out)
    out += contents + b'\x00'
    out += struct.pack('<I', mask_crc(crc32c(contents + b'\x00')))
    return _put_varint(offset) + _put_varint(len(contents))


def write_table(path, pairs, block_size=4096):
    """pairs: (key bytes, value bytes) in strictly increasing key order."""
    out, index, blk, last_key = bytearray(), _BlockBuilder(1), _BlockBuilder(), None
    for key, value in pairs:
        if last_key is not None and key <= last_key:
            raise BundleError('table keys must be strictly increasing')
        blk.add(key, value)
        last_key = key
        if len(blk.buf) >= block_size:
            index.add(key, _emit_block(out, blk.finish()))
            blk = _BlockBuilder()
    if blk.count:
        index.add(last_key, _emit_block(out, blk.finish()))
    meta = _emit_block(out, _BlockBuilder().finish())
    idx = _emit_block(out, index.finish())
    footer = meta + idx
    out += footer + b'\x00' * (40 - len(footer)) + struct.pack('<Q', TABLE_MAGIC)
    with open(path, 'wb') as f:
        f.write(out)


def write_bundle(prefix, tensors, block_size=4096):
    """{name: ndarray} -> prefix.index + prefix.data-00000-of-00001 (one shard, little endian)."""
    header = _field(1, 0, _put_varint(1)) + _field(3, 2, b'\x02' + _field(1, 0, _put_varint(1)))
    pairs, offset = [(b'', header)], 0
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    with open(prefix + '.data-00000-of-00001', 'wb') as f:
        for name in sorted(tensors, key=lambda s: s.encode('utf-8')):
            arr = np.asarray(tensors[name])
            if not arr.flags.c_contiguous:   # (ascontiguousarray would turn a scalar into shape (1,))
                arr = arr.copy(order='C')
            raw = arr.tobytes()
            f.write(raw)
            pairs.append((name.encode('utf-8'), _encode_entry(arr, offset, mask_crc(crc32c(raw)))))
            offset += len(raw)
    write_table(prefix + '.index', pairs, block_size)
