"""Is some region of the 288 GB slower than the rest?  34 buffers of 8 GiB allocated one after the other (all alive), each timed with
a write-only fill, a read-only sum and a copy between its halves (GB/s of bytes moved)."""
import torch
dev = torch.device("cuda", 0)
n = 1 << 30   # doubles = 8 GiB
bufs = []
for k in range(34):
    try:
        bufs.append(torch.empty(n, dtype=torch.float64, device=dev))
    except Exception as e:
        print("stopped at", k, e); break


def t(fn, reps=4):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


for k, b in enumerate(bufs):
    h = n // 2
    tf = t(lambda: b.fill_(1.0))
    tc = t(lambda: b[:h].copy_(b[h:]))
    tr = t(lambda: b.sum())
    print(f"buffer {k:2d} at {b.data_ptr():#x}: fill {8 * n / tf / 1e9:6.0f} GB/s   copy {8 * n / tc / 1e9:6.0f} GB/s   read {8 * n / tr / 1e9:6.0f} GB/s", flush=True)
