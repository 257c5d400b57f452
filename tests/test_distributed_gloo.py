"""N > 1 path on CPU: two gloo ranks shard the images, replay the reference's sequential RNG stream
and meet in the single all-gather of IoU records (the RCCL collective of the GPU run)."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _expected_record(g):
    return np.array([g + 0.1, g + 0.2, g + 0.3, g + 0.4, g + 0.5, g + 0.6]) / 100.0


def _worker(rank, world, port, num_images, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from asr_amd import distributed as D
    r, w, _ = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    mine = D.shard_indices(num_images, rank, world)
    params = D.replay_augmentation_stream(num_images, 5, 0.15, 80, seed=1234)
    recs = [_expected_record(g) for g in mine]
    table = D.all_gather_iou(mine, recs, num_images, device=torch.device("cpu"))
    q.put((rank, mine, table, [params[g][0].copy() for g in mine]))
    torch.distributed.destroy_process_group()


def test_two_rank_shard_and_allgather():
    world, num_images = 2, 7                       # ragged: rank 0 owns 4 images, rank 1 owns 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, num_images, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from asr_amd import distributed as D
    from asr_amd.superresolution_scripts.augmentation_utils import draw_augmentation_parameters
    exp = np.stack([_expected_record(g) for g in range(num_images)])
    np.random.seed(1234)
    seq = [draw_augmentation_parameters(5, 0.15, 80)[0] for _ in range(num_images)]   # the sequential reference stream
    owned = []
    for rank, mine, table, angles in results:
        assert mine == list(range(rank, num_images, world))
        np.testing.assert_array_equal(table, exp)                 # every rank holds the full table
        for g, a in zip(mine, angles):
            np.testing.assert_array_equal(a, seq[g])              # identical augmentation seeds under sharding
        owned += mine
    assert sorted(owned) == list(range(num_images))
    means = D.mean_ious(exp)
    assert abs(means["aug_single"] - exp[:, 2].mean()) < 1e-15


def test_single_process_paths_and_adam_counter():
    from asr_amd import distributed as D
    t = D.all_gather_iou([2, 0], [_expected_record(2), _expected_record(0)], 3)
    assert np.isnan(t[1]).all() and np.array_equal(t[0], _expected_record(0))
    assert D.adam_start_step(3, 300) == 900 and D.adam_start_step(3, 300, "slice_max") == 1800
    assert D.shard_indices(10, 3, 8) == [3] and D.shard_indices(0, 0, 2) == []
    state = np.random.get_state()[1].copy()
    D.replay_augmentation_stream(2, 4, 0.1, 10)
    assert np.array_equal(np.random.get_state()[1], state)        # the caller's global RNG is untouched


def _eval_worker(rank, world, port, root, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import sys
    import types
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from asr_amd import distributed as D, evaluation as E
    D.init_from_env(backend="gloo")
    # the device work is stubbed out: what is under test is the host logic around it -- which files a rank loads, the
    # validity exchange before the solves, and the Adam start step each image gets
    opened = []
    real_load = E.load_SR_data
    E.load_SR_data = lambda p, **kw: (opened.append(os.path.basename(p)), real_load(p, **kw))[1]
    E.compute_SR = lambda sr, *a, **kw: sr.optimizer.optimizer.iterations
    E.compute_IoU = lambda true, pred, **kw: float(pred)
    E.load_image = lambda *a, **kw: None
    sr = types.SimpleNamespace(num_iter=10, optimizer=types.SimpleNamespace(optimizer=types.SimpleNamespace(iterations=0)))
    paths = E.interchange_files(root)
    table, valid = E.evaluate_precomputed(sr, paths, root, num_aug=6, rank=rank, world=world)
    q.put((rank, [os.path.basename(p) for p in paths], opened, table, valid))
    import torch.distributed
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def test_two_rank_evaluation_validity_and_adam_start(tmp_path):
    """evaluation.evaluate_precomputed on two ranks and five files, one unreadable, one too short, one slice_max (two
    solves): each rank opens only its own shard, validity is exchanged before any solve, and every valid image starts
    its Adam counter where the reference's sequential loop would be (SR_single_class.py:83-90; optimizer.py's global
    step) -- the table equals the one-rank table."""
    from asr_amd.superresolution_scripts import superres_utils as su
    masks = np.zeros((6, 4, 4, 1), np.float32)
    a, sh = np.arange(6, dtype=np.float32), np.ones((6, 2), np.float32)
    su.save_SR_data(str(tmp_path / "1"), masks, None, a, sh, "1", "argmax", 0.15, 80)
    su.save_SR_data(str(tmp_path / "2"), masks, masks, a, sh, "2", "slice_max", 0.15, 80)          # two solves
    (tmp_path / "3.hdf5").write_bytes(b"not an hdf5 file")
    su.save_SR_data(str(tmp_path / "4"), masks[:3], None, a[:3], sh[:3], "4", "argmax", 0.15, 80)   # too few copies
    su.save_SR_data(str(tmp_path / "5"), masks, None, a, sh, "5", "argmax", 0.15, 80)
    expect_valid = [True, True, False, False, True]
    expect_start = [0, 10, None, None, 30]                 # num_iter x solves of the valid files before

    def run(world):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_eval_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
        for p in procs:
            p.start()
        res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        return res

    two, one = run(2), run(1)
    for rank, names, opened, table, valid in two + one:
        assert names == ["1.hdf5", "2.hdf5", "3.hdf5", "4.hdf5", "5.hdf5"]
        assert list(valid) == expect_valid
        for g, start in enumerate(expect_start):
            if start is None:
                assert np.isnan(table[g]).all()
            else:
                assert np.isnan(table[g, :2]).all() and list(table[g, 2:]) == [float(start)] * 4
    # own shard only, once, and only the VALID files (validity comes from the headers: probe_SR_data reads no mask)
    assert two[0][2] == ["1.hdf5", "5.hdf5"] and two[1][2] == ["2.hdf5"]
    np.testing.assert_array_equal(two[0][3], one[0][3])
