"""TensorFlow tensor-bundle checkpoints (host-side data format next to the hot path): the reader is
pinned by the format's published constants and by round trips through the writer; no checkpoint
written by TensorFlow is available, so parity with TensorFlow's own files is unpinned."""
import os
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'style-transfer_amd'), os.path.join(ROOT, 'style-transfer_amd', 'src')]
import tf_bundle as B


def test_crc32c_known_answers():
    assert B.crc32c(b'123456789') == 0xE3069283          # the Castagnoli check value
    assert B.crc32c(b'\x00' * 32) == 0x8A9136AA           # RFC 3720 B.4 test patterns
    assert B.crc32c(b'\xff' * 32) == 0x62A8AB43
    assert B.crc32c(bytes(range(32))) == 0x46DD794E
    for c in (0, 1, 0xE3069283, 0xffffffff):
        assert B.unmask_crc(B.mask_crc(c)) == c
    assert B.mask_crc(0) == 0xa282ead8


def test_table_round_trip_with_prefix_compression_and_many_blocks(tmp_path):
    pairs = [(b'', b'header')] + [(('style_network/Variable_%03d' % i).encode(), os.urandom(1 + i % 40)) for i in range(300)]
    pairs.sort()
    path = str(tmp_path / 't.index')
    B.write_table(path, pairs, block_size=256)
    assert B.read_table(path) == pairs
    data = open(path, 'rb').read()
    assert struct.unpack('<Q', data[-8:])[0] == 0xdb4775248b80fb57 and len(data) > 48
    # a flipped byte inside a block is caught by the block checksum
    bad = bytearray(data)
    bad[10] ^= 0x40
    open(path, 'wb').write(bytes(bad))
    with pytest.raises(B.BundleError):
        B.read_table(path)


def test_bundle_round_trip_and_errors(tmp_path):
    rng = np.random.default_rng(3)
    tensors = {'style_network/Variable': rng.standard_normal((9, 9, 3, 32)).astype(np.float32),
               'style_network/Variable_1': np.zeros(32, np.float32),
               'style_network/Variable_2': np.ones(32, np.float32),
               'style_network/Variable/Adam': rng.standard_normal((9, 9, 3, 32)).astype(np.float32),
               'beta1_power': np.float32(0.9), 'global_step': np.int64(7)}
    prefix = str(tmp_path / 'saved_m-0')
    B.write_bundle(prefix, tensors)
    got = B.read_bundle(prefix, verify_data=True)
    assert set(got) == set(tensors)
    for k, v in tensors.items():
        assert got[k].dtype == np.asarray(v).dtype and got[k].shape == np.asarray(v).shape
        assert np.array_equal(got[k], v)
    listed = B.list_bundle(prefix)
    assert listed['style_network/Variable'] == (np.dtype(np.float32), (9, 9, 3, 32)) and listed['beta1_power'][1] == ()
    with pytest.raises(KeyError):
        B.read_bundle(prefix, ['style_network/Variable_9'])
    # corrupt tensor bytes: caught when the data checksum is verified
    data_path = prefix + '.data-00000-of-00001'
    raw = bytearray(open(data_path, 'rb').read())
    raw[100] ^= 1
    open(data_path, 'wb').write(bytes(raw))
    with pytest.raises(B.BundleError):
        B.read_bundle(prefix, ['style_network/Variable'], verify_data=True)


def test_service_loader_reads_a_saver_checkpoint(tmp_path):
    """48 variables saved the way the reference's Saver names them, plus optimizer slots, load in
    creation order through the same function the service uses."""
    import full_graph_stylizer_network as F
    from stmi.stylizer import init_params
    params = init_params(seed=5)
    tensors = {F._var_key(i): np.asarray(p, np.float32) for i, p in enumerate(params)}
    tensors.update({F._var_key(i) + '/Adam': np.zeros_like(v) for i, v in enumerate(tensors.values())})
    tensors['beta1_power'] = np.float32(0.5)
    model_dir = tmp_path / 'STYLE_NET_x'
    model_dir.mkdir()
    B.write_bundle(str(model_dir / 'saved_STYLE_NET_x-0'), tensors, block_size=512)
    (model_dir / 'checkpoint').write_text('model_checkpoint_path: "saved_STYLE_NET_x-0"\n')
    path = F.latest_checkpoint(str(model_dir))
    assert path.endswith('saved_STYLE_NET_x-0.index')
    loaded = F.load_checkpoint(path)
    assert len(loaded) == len(params) == 48
    for a, b in zip(loaded, params):
        assert np.array_equal(a, np.asarray(b, np.float32))
