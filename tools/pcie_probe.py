import torch, time
n = 256 * 1024 * 1024
h_in = torch.empty(n, dtype=torch.uint8).pin_memory()
h_out = torch.empty(n, dtype=torch.uint8).pin_memory()
d_a = torch.empty(n, dtype=torch.uint8, device="cuda")
d_b = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(h2d, d2h, reps=8):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        if h2d:
            with torch.cuda.stream(s1): d_a.copy_(h_in, non_blocking=True)
        if d2h:
            with torch.cuda.stream(s2): h_out.copy_(d_b, non_blocking=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    return reps * n * (int(h2d) + int(d2h)) / dt / 1e9
run(True, True, 2)
print("H2D alone %.1f GB/s, D2H alone %.1f GB/s, both at once %.1f GB/s total" % (run(True, False), run(False, True), run(True, True)))
