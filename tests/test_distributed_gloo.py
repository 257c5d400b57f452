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
