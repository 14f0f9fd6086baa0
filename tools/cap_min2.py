import sys, torch
v = sys.argv[1]
dev = torch.device("cuda:0")
a = torch.zeros(1024, device=dev); b = torch.zeros(1024, device=dev); c = torch.zeros(1024, device=dev)
sa = torch.cuda.Stream(device=dev); side = [torch.cuda.Stream(device=dev) for _ in range(4)]
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    main = torch.cuda.current_stream()
    a.add_(1)
    sa.wait_stream(main)
    if v == "prefork":
        for i in range(4):
            side[i].wait_stream(main)
    with torch.cuda.stream(sa):
        b.add_(1)
        joins = []
        for i in range(4):
            ev = torch.cuda.Event(); ev.record(sa)
            b.add_(1)
            side[i].wait_event(ev)
            with torch.cuda.stream(side[i]):
                c.add_(1)
                e = torch.cuda.Event(); e.record(side[i]); joins.append(e)
        for e in joins[1:]:
            sa.wait_event(e)
        if v == "late0":
            pass
        else:
            sa.wait_event(joins[0])
    a.add_(1)
    if v == "late0":
        main.wait_event(joins[0])
    main.wait_stream(sa)
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
print(v, "ok", float(a[0]), float(b[0]), float(c[0]))
