import sys, torch
v = sys.argv[1]
dev = torch.device("cuda:0")
a = torch.zeros(1024, device=dev); b = torch.zeros(1024, device=dev); c = torch.zeros(1024, device=dev)
sa = torch.cuda.Stream(device=dev); sb = torch.cuda.Stream(device=dev)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    main = torch.cuda.current_stream()
    a.add_(1)
    sa.wait_stream(main)
    with torch.cuda.stream(sa):
        b.add_(1)
        ev = torch.cuda.Event(); ev.record(sa)
        sb.wait_event(ev)
        with torch.cuda.stream(sb):
            c.add_(1)
            e = torch.cuda.Event(); e.record(sb)
        if v == "joinsa":
            sa.wait_event(e)
    a.add_(1)
    if v != "joinsa":
        main.wait_event(e)
    main.wait_stream(sa)
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
print(v, "ok", float(a[0]), float(b[0]), float(c[0]))
