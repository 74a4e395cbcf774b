"""Fork/join edges of a captured graph under the M1 step's topology: main writes e before the fork, the lane reads it late; main keeps
working next to the lane (a result consumed after the join); main reads the lane's result after the join.  e changes every replay (a
counter kernel inside the graph): a stale read on either side shows as a wrong sum."""
import sys, torch
dev = torch.device("cuda:0")
N = 1 << 16
big = torch.randn(2048, 2048, device=dev)
def run(pre_join_work, nside):
    main, lane = torch.cuda.Stream(), torch.cuda.Stream()
    sides = [torch.cuda.Stream() for _ in range(nside)]
    cnt = torch.zeros(1, device=dev); e = torch.zeros(N, device=dev); z = torch.zeros(N, device=dev); c = torch.zeros(N, device=dev)
    out = torch.zeros(N, device=dev); tmp = torch.empty(2048, 2048, device=dev); tmp2 = torch.empty(2048, 2048, device=dev)
    sacc = [torch.zeros(N, device=dev) for _ in range(nside)]
    def body():
        cur = torch.cuda.current_stream()
        cnt.add_(1.0); e.copy_(cnt.expand(N))                    # e = replay index (before the fork)
        lane.wait_stream(cur)
        with torch.cuda.stream(lane):
            torch.mm(big, big, out=tmp)                          # the lane is busy for a while ...
            z.copy_(e); z.mul_(2.0)                              # ... then reads e
        if pre_join_work:
            torch.mm(big, big, out=tmp2); c.copy_(e); c.add_(0.5)   # main next to the lane; c is consumed after the join
        else:
            c.copy_(e); c.add_(0.5)
        for s, a in zip(sides, sacc):                            # side branches forked before the join, joined after it
            s.wait_stream(cur)
            with torch.cuda.stream(s): a.copy_(e); a.add_(1.0)
        cur.wait_stream(lane)                                    # join
        out.copy_(z); out.add_(c)
        for s, a in zip(sides, sacc):
            cur.wait_stream(s); out.add_(a)
    with torch.cuda.stream(main):
        body(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=main):
            body()
        bad = 0
        for r in range(300):
            g.replay(); torch.cuda.synchronize()
            k = float(cnt)
            want = 2 * k + k + 0.5 + nside * (k + 1.0)
            bad += int(not bool((out == want).all()))
    print(f"pre_join_work={pre_join_work} side branches={nside}: {bad} of 300 replays wrong")
for pj in (0, 1):
    for ns in (0, 3):
        run(pj, ns)
