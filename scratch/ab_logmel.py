"""logmel_kernel of this tree against the round-4 tree's (scratch/_r04_tree, see ab_rounds.sh): same bits, microseconds per 64 x 15 s batch.
Both libraries are loaded side by side through ctypes; the call is convasr_logmel_fwd of include/convasr_hip.h."""
import ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import convasr_amd as ca

d = torch.device('cuda:0')
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = dict(r04 = os.path.join(root, 'scratch/_r04_tree/convasr_amd/libconvasr_hip.so'), now = os.path.join(root, 'convasr_amd/libconvasr_hip.so'))
fe = ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window').to(d)
out = {}
res = {}
for B, secs, dt in ((64, 15, torch.float32), (32, 20, torch.int16), (3, 0.4, torch.float32)):
	T = int(16000 * secs)
	torch.manual_seed(1)
	x = torch.rand(B, T, device = d) * 2 - 1
	if dt == torch.int16: x = (x * 30000).to(torch.int16)
	xlen = torch.linspace(0.5, 1, B, device = d)
	absmax = x.float().abs().amax(1).contiguous()
	F = 1 + T // 160
	for name, path in libs.items():
		if not os.path.exists(path): continue
		lib = ctypes.CDLL(path)
		f = lib.convasr_logmel_fwd
		f.restype = ctypes.c_int
		f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_float, ctypes.c_void_p]
		y = torch.zeros(B, F, 64, device = d)
		args = (x.data_ptr(), 0 if dt == torch.float32 else 2, absmax.data_ptr(), xlen.data_ptr(), fe.window.data_ptr(), fe.window.shape[0], fe.mel.weight.data_ptr(), fe.mel.bias.data_ptr(), y.data_ptr(), B, T, 512, 160, 64, 0.97, torch.cuda.current_stream().cuda_stream)
		assert f(*args) == 0
		torch.cuda.synchronize()
		e0, e1 = torch.cuda.Event(enable_timing = True), torch.cuda.Event(enable_timing = True)
		for _ in range(5): f(*args)
		e0.record()
		for _ in range(50): f(*args)
		e1.record(); torch.cuda.synchronize()
		res[name] = y
		out[f'{B}x{secs}s {str(dt)[6:]} {name} us'] = round(e0.elapsed_time(e1) * 1000 / 50, 1)
	if len(res) == 2: out[f'{B}x{secs}s {str(dt)[6:]} bit-identical'] = bool(torch.equal(res['r04'], res['now']))
print(json.dumps(out, indent = 1))

# the other FFT sizes / channel counts of this tree (no round-4 counterpart: nfft 512 and 64 channels only there)
other = {}
for sr, wsize, nmel in ((8000, 0.02, 64), (8000, 0.01, 64), (16000, 0.04, 64), (44100, 0.02, 64), (16000, 0.025, 80), (16000, 0.02, 128)):
	f2 = ca.models.LogFilterBankFrontend(nmel, sr, wsize, 0.01, 'hann_window').to(d)
	x = torch.rand(64, 15 * sr, device = d) * 2 - 1
	xl = torch.ones(64, device = d)
	for _ in range(3): f2(x, xlen = xl)
	e0, e1 = torch.cuda.Event(enable_timing = True), torch.cuda.Event(enable_timing = True)
	e0.record()
	for _ in range(20): f2(x, xlen = xl)
	e1.record(); torch.cuda.synchronize()
	other[f'64x15s {sr} Hz window {wsize} s nfft {f2.nfft} mels {nmel}: us incl. absmax'] = round(e0.elapsed_time(e1) * 1000 / 20, 1)
print(json.dumps(other, indent = 1))
