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
