"""CPU checks of the implicit-GEMM 3x3 convolution (go_slam_amd/csrc/conv3x3_pp.hip): the host-side weight images, and
the kernel's schedule, per-thread index arithmetic + LDS bank behaviour through the lane-level emulator
(tools/emulate_conv3x3_pp.py).
The numerics on hardware are covered by tests/test_widen_gpu.py (-m gpu)."""
import torch


def _pp_emulator():
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "emulate_conv3x3_pp.py")
    spec = importlib.util.spec_from_file_location("emulate_conv3x3_pp", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_conv3x3_weight_image_layout():
    """pack_conv3x3_weight produces the LDS images gs_conv3x3_pp documents (include/goslam_hip.h): contracting them
    with the zero-padded input exactly the way the kernel indexes them reproduces F.conv2d."""
    import torch.nn.functional as F
    from go_slam_amd.droid_net import pack_conv3x3_weight
    g = torch.Generator().manual_seed(9)
    O, C, H, W = 256, 64, 5, 7
    w = (torch.randn(O, C, 3, 3, generator=g) * 0.1).half()
    x = torch.randn(1, C, H, W, generator=g).half()
    xp = F.pad(x.float(), (1, 1, 1, 1))[0].permute(1, 2, 0)                     # [H+2, W+2, C]
    ref = F.conv2d(x.float(), w.float(), padding=1)[0]
    for kc in (32,):
        wp = pack_conv3x3_weight(w, kc).float().view(O // 128, C // kc, 9, kc // 8, 128, 8)
        out = torch.zeros(O, H, W)
        for ck in range(C // kc):
            for tap in range(9):
                ky, kx = tap // 3, tap % 3
                win = xp[ky:ky + H, kx:kx + W, kc * ck:kc * ck + kc].reshape(H, W, kc // 8, 8)   # [.., kg, e]
                out += torch.einsum("nkre,hwke->nrhw", wp[:, ck, tap], win).reshape(O, H, W)
        assert torch.allclose(out, ref, rtol=1e-4, atol=1e-4), kc
    # 64 output channels (flow_encoder[2]): images of 64-channel blocks, [1][C/32][9][4][64][8]
    w64 = w[:64].contiguous()
    wp = pack_conv3x3_weight(w64, 32).float().view(1, C // 32, 9, 4, 64, 8)
    out = torch.zeros(64, H, W)
    for ck in range(C // 32):
        for tap in range(9):
            ky, kx = tap // 3, tap % 3
            win = xp[ky:ky + H, kx:kx + W, 32 * ck:32 * ck + 32].reshape(H, W, 4, 8)
            out += torch.einsum("nkre,hwke->nrhw", wp[:, ck, tap], win).reshape(64, H, W)
    assert torch.allclose(out, ref[:64], rtol=1e-4, atol=1e-4)


def test_conv3x3_xcd_aware_block_order_is_a_bijection_and_colocates_channel_blocks():
    """decode_block (conv3x3_common.h): every (tile, channel block) pair is produced exactly once for
    any tile count, and with the XCD-aware order the channel blocks of a tile get workgroup ids that are equal modulo 8
    (same XCD, hence one L2 for the tile's input patch) and at most 8 * (NB - 1) apart."""
    emu = _pp_emulator()
    for ntiles in list(range(1, 20)) + [1500, 1501, 1507]:
        for NB in (1, 2, 3):
            for xcd in (0, 1):
                got = [emu.decode_block(L, ntiles, NB, xcd) for L in range(ntiles * NB)]
                assert sorted(got) == [(t, b) for t in range(ntiles) for b in range(NB)], (ntiles, NB, xcd)
                if xcd:
                    ids = {}
                    for L, (t, b) in enumerate(got):
                        ids.setdefault(t, []).append(L)
                    full = (ntiles // 8) * 8
                    for t, ls in ids.items():
                        if t < full:
                            assert len({L % 8 for L in ls}) == 1 and max(ls) - min(ls) == 8 * (NB - 1), (ntiles, NB, t)


def test_conv7x7_c4_weight_image_by_lane_level_emulation():
    """pack_conv7x7_c4_weight vs conv7x7.hip's indexing, lane by lane: A fragment (nh, t, s) lane l = 8 halves of channel
    64 nh + 32 t + (l & 31) at K = 16 s + 8 (l >> 5) ..; the pixel operand of lane l at k-step s = the 2 patch pixels
    (py + s // 2, px + 4 (s & 1) + 2 (l >> 5) + {0, 1}) x 4 channels of the zero-padded NHWC patch (origin -3, -3).
    Contracting the two the way the MFMA does reproduces F.conv2d; the 8th tap never contributes."""
    import torch.nn.functional as F
    from go_slam_amd.droid_net import pack_conv7x7_c4_weight
    g = torch.Generator().manual_seed(4)
    H, W = 5, 9
    w = (torch.randn(128, 4, 7, 7, generator=g) * 0.1).half()
    x = torch.randn(1, 4, H, W, generator=g).half()
    ref = F.conv2d(x.float(), w.float(), padding=3)[0]                          # [128, H, W]
    img = pack_conv7x7_c4_weight(w).float().view(2, 2, 14, 64, 8)
    patch = torch.full((H + 6, W + 8, 4), float("nan"))                         # kernel: zero-filled incl. columns W+3..W+4
    patch[:] = 0.0
    patch[3:3 + H, 3:3 + W] = x[0].permute(1, 2, 0).float()
    out = torch.zeros(128, H, W)
    for py in range(H):
        for px in range(W):
            for s in range(14):
                for kg in range(2):
                    pix = patch[py + s // 2, px + 4 * (s & 1) + 2 * kg: px + 4 * (s & 1) + 2 * kg + 2].reshape(8)
                    lanes = torch.arange(32) + 32 * kg
                    a = img[:, :, s, lanes]                                     # [nh, t, 32 rows, 8]
                    out[:, py, px] += (a @ pix).reshape(128)
    assert torch.allclose(out, ref, rtol=1e-4, atol=1e-4)
    wk = img.permute(0, 1, 3, 2, 4)                                              # nh t lane s e
    # tap 7 of every kernel row (K = 32 ky + 28 .. 31 -> k-step 2 ky + 1, k-group 1, e = 4..7) is zero
    assert float(img[:, :, 1::2, 32:, 4:].abs().max()) == 0.0 and wk.shape[-2] == 14


def test_pingpong_kernel_schedule_and_data_path_by_emulation():
    """tools/emulate_conv3x3_pp.py restates conv3x3_pp_kernel (the production convolution): (1) the LDS-DMA / counted-wait /
    barrier schedule of both wave groups -- every fragment read must find its tap's weights and its chunk's patch
    published by BOTH groups, no buffer refilled before its last reader -- for 1, 2, 4 and 10 channel chunks and both
    tile widths; (2) every thread's index arithmetic (DMA sources incl. zero page and swizzle, fragment slots incl. the
    image-boundary masks, MFMA operand layout, lane permutation, epilogue transpose) against F.conv2d on ragged shapes;
    (3) the ds_read_b128 bank model: every pixel-fragment read conflict-free."""
    E = _pp_emulator()
    for tw in (16, 8):
        assert set(E.bank_model(tw)) == {1}, "pixel-fragment reads must be conflict-free"
        for nchunk in (1, 2, 4, 10):
            problems = E.schedule(nchunk, tw)
            assert not problems, problems[:4]
    # (the last case is the 64-output-channel instantiation: 4 KB weight images, one 32-channel fragment per wave)
    for args in ((2, 5, 19, 32, 40, 128, 16, 0), (3, 7, 10, 64, 64, 128, 8, 1), (2, 9, 21, 64, 64, 64, 16, 1)):
        nan, err = E.check(*args)
        assert nan == 0 and err < 1e-4, (args, nan, err)
