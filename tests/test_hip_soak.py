"""A long-running server calls the engine millions of times: device memory must not creep.  Mixed calls in a loop;
after a warm-up round (pooled scratch grows once) the free device memory has to stay where it was."""
import numpy as np
import pytest

from conftest import ids_for

pytestmark = pytest.mark.gpu


def test_no_device_memory_growth_under_mixed_calls(hip, oracle):
    import torch
    n, d = 20000, 384
    rows = oracle.synth_rows(n, d)
    ids = ids_for(n)
    h = hip.HipIndex(d)
    h.insert_batch(ids, rows)
    qs = oracle.synth_queries(n, d, 96)
    extra = oracle.synth_rows(n + 4000, d)[n:]
    extra_ids = ids_for(n + 4000)[n:]

    def round_(r):
        h.search(qs[r % 96], 10, None)
        h.search(qs[(r + 1) % 96], 300, None)                       # sort path
        h.search_threshold(qs[r % 96], 0.8, None)
        h.search_batch_arrays(qs[:64], 10)
        h.search_batch_arrays(qs[:40], 100)                         # wide lists, 64-query mode
        h.search_batch_arrays(qs[:20], 100)                         # wide lists, 32-query mode
        h.search(qs[r % 96], 5, hip.VectorFilter(exclude=[ids[i].tobytes() for i in range(5)]))
        h.autolink_pass_rows(None, 100, 0.85, 50)
        h.autolink_pass_rows(np.arange(40, dtype=np.uint32), 100, 0.85, 50)
        h.topk_lists_rows(100, np.arange(70, dtype=np.uint32))
        j = (r * 13) % 4000
        h.insert(extra_ids[j].tobytes(), extra[j])                  # new row or in-place upsert
        h.remove(extra_ids[(j + 7) % 4000].tobytes())
        if r % 10 == 9:
            h.rebuild()

    for r in range(12):          # warm-up: scratch pools, derived copies, rebuild buffers reach their sizes
        round_(r)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for r in range(12, 72):
        round_(r)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    # rows added since the baseline account for < 4000 x 384 x 4 x (rows + split + shadow) ~ 15 MB; allow 64 MB in all
    assert free0 - free1 < 64 << 20, f"device memory shrank by {(free0 - free1) >> 20} MiB over 60 rounds"


def test_batched_search_scratch_does_not_scale_with_the_store(hip):
    """Round 3's screening pass kept candidate lists with room for every row: 512 bytes per ROW and context (5 GB per context at
    10M rows, times the rotating streams, times the pooled readers).  The lists are bounded now (CX_BATCHS_CAND_CAP, 65,536
    entries per query = 33.5 MB per context whatever the store's size; a list that runs over is redone exactly): the first
    batched search of a 2M-row store may take the shadow (rows x dim x 2) plus well under 0.1 GB — round 3 took 1 GB on top —,
    and four more contexts (four host threads at once) under 0.3 GB."""
    import threading
    import torch
    from cortex_amd import _lib
    L = _lib.load()
    n, d = 2_000_000, 128
    gen = torch.empty((n, d), dtype=torch.float32, device="cuda:0")
    assert L.cx_synth_fill_dev(0, gen.data_ptr(), 20260313, 20260313, 20260315, n // 50, 0, n, d, 1) == 0
    ids = np.zeros((n, 16), np.uint8)
    ids[:, 8:] = np.arange(n, dtype=np.uint64).astype(">u8").view(np.uint8).reshape(n, 8)
    h = hip.HipIndex(d)
    h.insert_batch_dev(ids, gen.data_ptr(), n, d)
    qs_t = torch.empty((64, d), dtype=torch.float32, device="cuda:0")
    assert L.cx_synth_fill_dev(0, qs_t.data_ptr(), 20260313, 20260314, 20260315, n // 50, 0, 64, d, 0) == 0
    qs = qs_t.cpu().numpy()
    h.search_arrays(qs[0], 10)                       # the scan's own scratch
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    h.search_batch_arrays(qs, 10)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    shadow = ((n + 255) // 256) * 256 * d * 2
    assert free0 - free1 < shadow + (100 << 20), f"first batched search took {(free0 - free1) >> 20} MiB, shadow {(shadow) >> 20} MiB"
    th = [threading.Thread(target=lambda: [h.search_batch_arrays(qs, 100) for _ in range(3)]) for _ in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    free2, _ = torch.cuda.mem_get_info()
    assert free1 - free2 < 300 << 20, f"four concurrent readers took {(free1 - free2) >> 20} MiB of scratch"
