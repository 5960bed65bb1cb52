"""BASELINE configs[1] (fp32 forward, 32 x 10 s) and the fused-eval inference path (f1), timed with the bench protocol."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import convasr_amd as ca
d = torch.device('cuda:0')
def timeit(fn, warm = 2, iters = 5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters
out = {}
for name, B, secs, dt, mode in [('config1_fp32_fwd_train_stats', 32, 10, torch.float32, 'train'), ('bf16_fwd_train_stats_64x15', 64, 15, torch.bfloat16, 'train'), ('bf16_eval_fused_64x15', 64, 15, torch.bfloat16, 'eval'), ('fp32_eval_fused_32x10', 32, 10, torch.float32, 'eval')]:
    torch.manual_seed(1)
    fe = ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window')
    model = ca.models.Wav2Letter(64, [38], frontend = fe, dropout = 0, check_time_dim_padded = False, compute_dtype = dt).to(d)
    x = (torch.rand(B, 16000 * secs) * 2 - 1).to(d); xlen = torch.ones(B, device = d)
    y = torch.randint(0, 37, (B, 1, 10 * secs), device = d); ylen = torch.full((B, 1), 10 * secs, device = d)
    if mode == 'train':
        model.train()
        def fn():
            with torch.no_grad(): return model(x, xlen, y = y, ylen = ylen)
    else:
        model.train()
        with torch.no_grad(): model(x, xlen)  # populate BN running stats
        model.eval(); model.fuse_conv_bn_eval()
        def fn():
            with torch.no_grad(): return model(x, xlen)
    s = timeit(fn)
    out[name] = dict(ms = round(s * 1e3, 2), audio_s_per_s = round(B * secs / s, 1), conv_tflops = round(6.67e9 * B * secs / s / 1e12, 1))
    print(name, out[name], flush = True)
    del model
os.makedirs('gpurun_out', exist_ok = True)
json.dump(out, open('gpurun_out/other_configs.json', 'w'), indent = 1)
