"""Full-size (BASELINE configs[1]: 512x512, num_aug=100, 128x128 features) checks of the HIP path through
size-independent properties -- the oracle is too slow at this size, so the kernels are held to identities
that must hold exactly or to rounding: identity copies, integer shifts, linearity of the warps, realign of
un-augmented copies == plain upsampling, mean of equal copies, batch invariance and determinism of the
model, threshold/IoU idempotence, SR invariants."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

H = W = 512
h = w = 128
N = 100


@pytest.fixture(scope="module")
def params():
    from asr_amd import distributed as D
    a, s = D.replay_augmentation_stream(1, N, 0.15, 80, seed=1234)[0]
    return a, s


def test_augment_identity_shift_and_linearity(dev, params):
    from asr_amd import ops
    from asr_amd.superresolution_scripts.augmentation_utils import augment_on_device
    angles, shifts = params
    g = torch.Generator(device="cpu").manual_seed(0)
    img = torch.rand((H, W, 3), generator=g).to(dev)
    img2 = torch.rand((H, W, 3), generator=g).to(dev)
    c1 = augment_on_device(img, angles, shifts)
    assert c1.shape == (N, H, W, 3)
    assert torch.equal(c1[0], img)                                     # copy 0 is never augmented
    # integer shift: exact pixel move with zero fill
    z = np.zeros(2, np.float32)
    c = augment_on_device(img, z, np.array([[0, 0], [37, -12]], np.float32))
    exp = torch.zeros_like(img)
    exp[:-12, 37:] = img[12:, :-37]
    assert torch.equal(c[1], exp)
    # linearity of both resamplings (rounding only)
    c2 = augment_on_device(img2, angles, shifts)
    c12 = augment_on_device(0.25 * img + 0.5 * img2, angles, shifts)
    assert (c12 - (0.25 * c1 + 0.5 * c2)).abs().max().item() < 5e-6
    # value range is preserved by convex interpolation + zero fill
    assert c1.min().item() >= 0.0 and c1.max().item() <= img.max().item() + 1e-6


def test_realign_of_unaugmented_copies_is_plain_upsampling(dev):
    from asr_amd import ops
    from asr_amd.superresolution_scripts.superresolution import Superresolution
    g = torch.Generator(device="cpu").manual_seed(1)
    y1 = torch.rand((h, w), generator=g).to(dev)
    y = y1[None, None].expand(1, N, h, w).contiguous()
    sr = Superresolution(1, 0, 0, 0, num_aug=N, feature_size=(h, w), output_size=(H, W))
    zeros_a, zeros_s = np.zeros((1, N), np.float32), np.zeros((1, N, 2), np.float32)
    up = ops.resize_bilinear(y1[None, :, :, None].expand(1, h, w, 4).contiguous(), (H, W))[0, :, :, 0]
    mx = sr.realign_batch(y, zeros_a, zeros_s, "max")[0]
    mn = sr.realign_batch(y, zeros_a, zeros_s, "mean")[0]
    x0 = ops.sr_init_target(y, (H, W))[0]                               # the solver's own half-pixel upsampling
    assert torch.equal(mx, x0)                                          # max of equal copies through identity warps: exact
    assert (mn - x0).abs().max().item() < 1e-5                          # mean = sum of 100 equal terms / 100
    assert (x0 - up).abs().max().item() < 1e-6                          # layer resize kernel (FMA-contracted) vs SR kernels


def test_sr_solver_invariants_full_size(dev, params):
    """All-zero masks stay exactly zero (gradient 0 -> Adam step 0); with only the data term and identical
    un-augmented copies of a constant, the residual is exactly zero and x does not move."""
    from asr_amd.superresolution_scripts.optimizer import Optimizer
    from asr_amd.superresolution_scripts.superresolution import Superresolution
    angles, shifts = params
    opt = Optimizer("adam", 1e-3, amsgrad=True, lr_scheduler=True, decay_steps=60, decay_rate=0.3)
    sr = Superresolution(1.0, 0.3, 0.7, 0.0, num_iter=10, num_aug=N, optimizer=opt, feature_size=(h, w),
                         output_size=(H, W))
    y0 = torch.zeros((1, N, h, w), device=dev)
    x, terms = sr.augmented_superresolution_batch(y0, angles[None], shifts[None])
    assert torch.count_nonzero(x).item() == 0 and float(terms.abs().max()) == 0.0
    assert opt.optimizer.iterations == 10
    sr2 = Superresolution(1.0, 0.0, 0.0, 0.0, num_iter=5, num_aug=N, optimizer=opt, feature_size=(h, w),
                          output_size=(H, W))
    yc = torch.full((1, N, h, w), 0.75, device=dev)
    x2, terms2 = sr2.augmented_superresolution_batch(yc, np.zeros((1, N), np.float32), np.zeros((1, N, 2), np.float32))
    assert torch.all(x2 == 0.75) and float(terms2[0, 0]) == 0.0


def test_model_batch_invariance_and_determinism(dev):
    """A copy's logits do not depend on its position in the batch or on the batch size (inference BN is folded,
    every kernel is per-pixel-row deterministic); two runs are bit-identical."""
    from asr_amd import weights as Wt
    from asr_amd.model import DeeplabModel
    model = DeeplabModel(Wt.make_synthetic_weights(1234), (H, W, 3), 21, False, None)
    g = torch.Generator(device="cpu").manual_seed(2)
    x = torch.rand((6, H, W, 3), generator=g).to(dev)
    a = model.predict_device(x, batch_size=6)
    b = model.predict_device(x, batch_size=6)
    assert a.shape == (6, h, w, 21) and torch.equal(a, b)
    c = model.predict_device(x.flip(0).contiguous(), batch_size=4).flip(0)
    assert torch.equal(a, c)
    assert torch.isfinite(a).all()


def test_opm_threshold_iou_properties_full_size(dev):
    from asr_amd import ops
    from asr_amd.utils import compute_IoU, create_mask
    g = torch.Generator(device="cpu").manual_seed(3)
    cls = torch.randint(0, 21, (N, h, w), generator=g)
    logits = torch.nn.functional.one_hot(cls, 21).float().to(dev) * 3.0 - 1.0
    assert torch.equal(ops.argmax(logits).cpu().long(), cls)
    m = ops.opm_argmax(logits, 8)
    assert torch.equal(m.cpu(), torch.where(cls == 8, 8.0, 0.0))
    assert create_mask(logits).shape == (N, h, w, 1)
    s, mx = ops.opm_slice_max(logits, 8)
    assert torch.equal(s.cpu(), torch.where(cls == 8, 2.0, -1.0)) and torch.equal(mx.cpu(), torch.where(cls == 8, -1.0, 2.0))
    sl = ops.opm_slice(logits, 8)
    assert torch.equal(sl.cpu(), torch.where(cls == 8, 1.0, 0.0))     # (v - min) / (max - min) with min=-1, max=2
    img = torch.rand((H, W), generator=g).to(dev)
    t1 = ops.threshold(img, 8, th_factor=0.2)
    t2 = ops.threshold(t1.float(), 8, th_factor=0.2)                   # thresholding a {0,8} mask is idempotent
    assert torch.equal(t1, t2)
    assert compute_IoU(t1, t1, img_size=(H, W), class_id=8) == 1.0
    inv = torch.where(t1 == 8, 0, 8).to(torch.int32)
    assert compute_IoU(t1, inv, img_size=(H, W), class_id=8) == 0.0
    # IoU is symmetric and equals |A n B| / |A u B| computed on the host
    other = ops.threshold(torch.rand((H, W), generator=g).to(dev), 8, th_factor=0.5)
    a, b = (t1 == 8).cpu().numpy(), (other == 8).cpu().numpy()
    ref = (a & b).sum() / (a | b).sum()
    assert compute_IoU(t1, other, img_size=(H, W), class_id=8) == ref == compute_IoU(other, t1, img_size=(H, W), class_id=8)


def _fused_entry_launches(model, dev):
    """The fused stem and the two fused separable convs of entry-flow block 1 of a 100-copy forward pass (name, args): MFMA kernels
    whose two waves per SIMD leave room for a co-resident wave of another stream -- the kernels beside which round 3's SR
    solves went wrong (packed-f32 with op_sel = [0,1] beside an MFMA in flight)."""
    g = torch.Generator(device="cpu").manual_seed(5)
    xin = torch.rand((N, H, W, 3), generator=g).to(dev)
    model.engine.forward(xin, lane=0)
    torch.cuda.synchronize()
    plan = model.engine.plan(N, H, W, 0)
    fused = [(name, args) for name, args, *_ in plan["steps"] if name in ("asr_entry_stem_f16x3", "asr_sepconv_fused_f16x3")]
    assert [n for n, _ in fused] == ["asr_entry_stem_f16x3", "asr_sepconv_fused_f16x3", "asr_sepconv_fused_f16x3"]
    return fused


def _replay(lib, launches, stream, times):
    from asr_amd import _lib
    with torch.cuda.stream(stream):
        s = _lib.stream_ptr()
        for _ in range(times):
            for name, args in launches:
                assert getattr(lib, name)(*args, s) == 0


def test_sr_solve_next_to_the_fused_entry_kernels_on_another_stream(dev):
    """An SR iteration that runs while entry_stem_fused_kernel / sepconv_fused_kernel occupy the chip on ANOTHER stream
    returns the values of a quiet run, stage by stage (residuals, gradient planes, x).  Round 3 found K_fwd returning
    garbage in lanes 48-63 of a few waves here (12 of 12 trials) while sr.hip was compiled with packed-f32 instructions;
    round 4 (profiles/r04_hazard_matrix.txt) reduced it to an instruction pair -- a packed-f32 instruction with op_sel = [0,1]
    on a vector src1, beside an MFMA of another wave on the same SIMD -- and csrc/isa_guard.py keeps the form out of every kernel
    (tools/diag_sr_stages_under_stem.py is the long form of this test)."""
    from asr_amd import _lib, ops, transforms as T, weights as Wt
    from asr_amd.model import DeeplabModel
    model = DeeplabModel(Wt.make_synthetic_weights(1234), (H, W, 3), 21, False, None)
    fused = _fused_entry_launches(model, dev)
    lib = _lib.load()

    rng = np.random.RandomState(3)
    y = ops.to_device((rng.rand(1, N, h, w) > 0.6).astype(np.float32))
    angles = rng.uniform(-0.15, 0.15, N).astype(np.float32)
    shifts = rng.uniform(-80, 80, (N, 2)).astype(np.float32)
    tf = lambda a: ops.to_device(a.reshape(1, N, 8))
    rot, irot = tf(T.rotation_transforms(angles, H, W)), tf(T.rotation_transforms(-angles, H, W))
    tr, itr = tf(T.translation_transforms(shifts)), tf(T.translation_transforms(-shifts))
    b1, b2 = np.float32(0.9), np.float32(0.999)
    alphas = ops.to_device(np.array([[T.adam_alpha(np.float32(1e-3), b1, b2, 1)]], np.float32))
    cfg = ops.sr_config(_lib.OPT_ADAM, True, np.float32(1) - b1, np.float32(1) - b2, np.float32(1e-7))

    def one_iteration():
        st = {}
        x0 = ops.sr_init_target(y, (H, W))
        ops.sr_solve(x0, y, rot, tr, irot, itr, alphas, (1.0, 0.3, 0.7, 0.0), want_loss=False, cfg=cfg, state=st)
        return x0, st["ws"]

    def stages(x, ws):      # sr.hip, asr_sr_solve_cfg_f32: resid | x_alt | acc | planes | bordered x | flags
        pe = (H + 4) * (W + 64)
        o = N * h * w + 2 * H * W
        chunk = (ws.numel() - 1 - o - pe) // pe
        return ws[:N * h * w], ws[o:o + chunk * pe], x

    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(sb):
        xq, wsq = one_iteration()
    torch.cuda.synchronize()
    quiet = [t.clone() for t in stages(xq, wsq)]
    for aggressors, times in ((fused[:1], 6), (fused[1:], 3)):            # the stem alone (0.9 ms each), the two separable convs
        for trial in range(6):
            _replay(lib, aggressors, sa, times)
            with torch.cuda.stream(sb):
                xt, wst = one_iteration()
            torch.cuda.synchronize()
            for what, got, ref in zip(("residuals", "gradient planes", "x"), stages(xt, wst), quiet):
                assert torch.equal(got, ref), (f"{aggressors[0][0]} trial {trial}: {what} differ in "
                                               f"{int((got != ref).sum())} elements")


def test_second_lane_kernels_next_to_the_fused_entry_kernels_on_another_stream(dev):
    """The OTHER kernels of the second lane -- the augmentation warp, the max / mean realignment, OPM argmax, min-max, the
    threshold and the IoU counts -- next to the same launches: bit-identical to their quiet runs."""
    from asr_amd import _lib, ops, transforms as T, weights as Wt
    from asr_amd.model import DeeplabModel
    model = DeeplabModel(Wt.make_synthetic_weights(1234), (H, W, 3), 21, False, None)
    fused = _fused_entry_launches(model, dev)
    lib = _lib.load()
    rng = np.random.RandomState(11)
    img = ops.to_device(rng.rand(H, W, 3).astype(np.float32))
    angles = rng.uniform(-0.15, 0.15, N).astype(np.float32)
    shifts = rng.uniform(-80, 80, (N, 2)).astype(np.float32)
    angles[0] = 0
    shifts[0] = 0
    rot, tr = ops.to_device(T.rotation_transforms(angles, H, W)), ops.to_device(T.translation_transforms(shifts))
    irot, itr = (ops.to_device(T.rotation_transforms(-angles, H, W)).reshape(1, N, 8),
                 ops.to_device(T.translation_transforms(-shifts)).reshape(1, N, 8))
    y = ops.to_device(rng.rand(1, N, h, w).astype(np.float32))
    logits = ops.to_device(rng.randn(N, h, w, 21).astype(np.float32))
    label = ops.to_device((rng.rand(H, W) > 0.5).astype(np.int32) * 8, dtype=torch.int32)

    def second_lane():
        copies = ops.augment_copies(img, rot, tr)
        mx, mean = ops.realign(y, itr, irot, (H, W), "both")
        masks = ops.opm_argmax(logits, 8)
        lo_hi = ops.minmax(mean)
        th = ops.threshold(mean.reshape(H, W), 8, th_factor=0.15)
        counts = ops.iou_counts(label, th, 8)
        return [copies, mx, mean, masks, lo_hi, th, counts]

    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(sb):
        quiet = second_lane()
    torch.cuda.synchronize()
    quiet = [torch.as_tensor(t).clone() if torch.is_tensor(t) else t for t in quiet]
    names = ("augment_copies", "realign max", "realign mean", "opm_argmax", "minmax", "threshold", "iou_counts")
    for trial in range(4):
        _replay(lib, fused, sa, 3)
        with torch.cuda.stream(sb):
            got = second_lane()
        torch.cuda.synchronize()
        for what, a, b in zip(names, got, quiet):
            same = torch.equal(a, b) if torch.is_tensor(a) else (np.asarray(a) == np.asarray(b)).all()
            assert same, f"trial {trial}: {what} differs from its quiet run"
