"""Does a forked stream keep its in-stream order inside a HIP graph capture after ANOTHER stream waited for it mid-way?
lane: K1 (slow, writes a) ... [main or a third stream waits for the lane] ... lane: K2 (reads a).  K2 depends on K1 through stream order only."""
import sys, torch
dev = torch.device("cuda:0")
N = 1 << 20
big = torch.randn(6144, 6144, device=dev)
def run(mode):
    main, lane, third = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    a = torch.zeros(N, device=dev); b = torch.zeros(N, device=dev); sink = torch.zeros(N, device=dev)
    tmp = torch.empty(6144, 6144, device=dev)
    def body():
        lane.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(lane):
            torch.mm(big, big, out=tmp)              # slow
            a.copy_(tmp.view(-1)[:N]); a.add_(1.0)   # K1
        if mode == "main":
            torch.cuda.current_stream().wait_stream(lane); sink.add_(1.0)
        elif mode == "third":
            third.wait_stream(torch.cuda.current_stream()); third.wait_stream(lane)
            with torch.cuda.stream(third): sink.add_(1.0)
        with torch.cuda.stream(lane):
            b.copy_(a); b.mul_(2.0)                  # K2: ordered behind K1 by the lane alone
        torch.cuda.current_stream().wait_stream(lane)
        if mode == "third": torch.cuda.current_stream().wait_stream(third)
    with torch.cuda.stream(main):
        body(); torch.cuda.synchronize()
        ref = b.clone()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=main):
            body()
        bad = 0
        for _ in range(10):
            a.zero_(); b.zero_(); torch.cuda.synchronize()
            g.replay(); torch.cuda.synchronize()
            bad += int(not torch.equal(b, ref))
    print(f"mode {mode}: {bad} of 10 replays differ from the eager result")
for m in ("none", "main", "third"):
    run(m)
