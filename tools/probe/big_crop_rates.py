import sys, os, time, torch
sys.path.insert(0, os.getcwd())
from doc2tex_amd import Model, synth
for (H, W, B) in ((448, 960, 16), (800, 800, 8)):
    cfg = synth.make_config("C2", device="cuda")
    cfg["max_dimension"] = [H, W]
    m = Model(cfg)
    m.load_state_dict(synth.synth_state_dict({k: v for k, v in m.state_dict().items()}), strict=False)
    m = m.cuda().eval()
    img = synth.synth_images(B, H, W, seed=5).cuda()
    text = torch.full((B, 1), 1, dtype=torch.long, device="cuda")
    with torch.no_grad():
        for _ in range(2):
            m.forward_encoder(img)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            m.forward_encoder(img)
        torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 3
        m(img, text, is_train=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m(img, text, is_train=False)
        torch.cuda.synchronize(); tf = time.perf_counter() - t0
    eng = m.engine(); eng.profile(True)
    with torch.no_grad():
        m.forward_encoder(img)
    torch.cuda.synchronize(); eng.profile(False)
    gemm = sum(r[3] for r in eng.profile_read(4096))
    print(f"{H}x{W} B={B}: T={m.engine().encoder_shape(H, W)[0]} encoder {te*1e3:.1f} ms (GEMM launches {gemm:.1f} ms), encoder + 151 greedy steps {tf*1e3:.1f} ms = {B/tf:.1f} formulas/s", flush=True)
    del m
