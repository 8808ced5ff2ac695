"""The C-ABI shared library loads without a GPU and exports every symbol include/mae_hip.h declares (no compute calls)."""
import ctypes
import re
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
HEADER = ROOT / "include" / "mae_hip.h"


def declared_functions():
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(mae_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from ssrl_vit_mae_jepa_amd import _lib
    names = declared_functions()
    assert len(names) >= 25
    exported = subprocess.run(["nm", "-D", "--defined-only", str(_lib.LIB_PATH)], capture_output=True, text=True, check=True).stdout
    for n in names:
        assert re.search(rf"\bT {n}\b", exported), f"{n} declared in mae_hip.h but not exported"
        assert hasattr(_lib.lib, n)
    assert sorted(_lib.SIGNATURES) == names  # the ctypes binding covers the header exactly


def test_abi_version_and_error_channel():
    from ssrl_vit_mae_jepa_amd import _lib
    assert _lib.lib.mae_abi_version() == _lib.ABI_VERSION == 4  # v2: image_dtype arguments (uint8 pixels); v3: mae_augment_crop_flip_u8; v4: the optimizer on a shard of the arena
    cfg = _lib.MaeConfig(image_size=96, patch_size=7, in_chans=3, embed_dim=384, depth=12, num_heads=6,
                         decoder_embed_dim=192, decoder_depth=2, decoder_num_heads=6, mlp_ratio=4, act_dtype=1)
    h = ctypes.c_void_p()
    rc = _lib.lib.mae_engine_create(ctypes.byref(cfg), ctypes.byref(h))
    assert rc != 0 and b"patch_size" in _lib.lib.mae_last_error()


def test_engine_metadata_without_gpu():
    from ssrl_vit_mae_jepa_amd.mae import Engine
    e = Engine(dict(image_size=96, patch_size=8, in_chans=3, embed_dim=384, depth=12, num_heads=6, decoder_embed_dim=192,
                    decoder_depth=2, decoder_num_heads=6), "bf16")
    assert sum(n for _, _, n, _, _ in e.table) == 22_454_016
    assert e.trainable_elems % 64 == 0 and e.arena_elems >= 22_454_016
    offs = sorted((o, n) for _, o, n, _, _ in e.table)
    assert all(o % 64 == 0 for o, _ in offs) and all(a[0] + a[1] <= b[0] for a, b in zip(offs, offs[1:]))
    ws_small, ws_big = e.workspace_bytes(64, 36), e.workspace_bytes(2000, 36)
    assert 0 < ws_small < ws_big < 64 * 2 ** 30
