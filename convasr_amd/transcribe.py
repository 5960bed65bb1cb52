"""MI355X-native mirror of the model-facing surface of the reference's transcribe.py.

* setup(args) -> (text_pipeline, frontend, model, generator)                          transcribe.py:23-60
  checkpoint -> frontend + model (the reference's class names / state-dict keys) -> eval() -> fuse_conv_bn_eval() ->
  compute dtype from args.fp16 (None / 'O0': exact fp32; 'O1'..'O3': fp16 storage + MFMA as under apex.amp, or bf16 when models.AMP_DTYPE says so) -> greedy generator.
* transcribe_batch(...)                                                                transcribe.py:140-200
  the per-batch body of transcribe.main: forward (every op a HIP kernel; the argmax of the greedy decode too), per-frame time
  stamps, GreedyCTCGenerator with time stamps (one segment per word), optional forced alignment of the reference text
  (ctc.alignment on the GPU) -> ref segments.

Out of scope: file discovery, audio decoding, the dataset, text pre/post-processing pipelines (regexes, number
normalisation), CER / error analysis and the json / html / csv / txt writers around this body.  A TextPipeline here is the tokenizer
plus identity pre/post-processing; pass the reference's own pipeline object as args.text_pipeline to get its text handling."""
import types

import torch

from . import ctc, models
from .transcript_generators import CharTokenizerLegacy, GreedyCTCGenerator

RU_ALPHABET = 'абвгдеёжзийклмнопрстуфхцчшщъыьэюя'  # configs/ru_text_config.json:10


class TextPipeline:
	"""What transcribe.py uses of text_processing.ProcessingPipeline: .tokenizer, .preprocess, .postprocess."""

	def __init__(self, tokenizer, preprocess = None, postprocess = None):
		self.tokenizer = tokenizer
		self.preprocess = preprocess or (lambda s: s)
		self.postprocess = postprocess or (lambda s: s)


def map_text(postprocess, hyp = (), ref = ()):
	"""transcripts.map_text (transcripts.py:34-35)."""
	return [dict(t, hyp = postprocess(t.get('hyp', ''))) for t in hyp] + [dict(t, ref = postprocess(t.get('ref', ''))) for t in ref]


def join(ref = (), hyp = ()):
	"""transcripts.join (transcripts.py:82-83)."""
	return ' '.join(filter(bool, [t.get('ref', '').strip() for t in ref] + [t.get('hyp', '').strip() for t in hyp]))


def setup(args):
	"""transcribe.setup(args).  args: namespace with checkpoint (a path, or the dict torch.load would return: 'args' and
	'model_state_dict'), device, fp16 (apex opt level or None), frontend_in_model, model (optional override), dither / dither0 /
	normalize_signal (optional), text_pipeline (optional; default: the legacy 38-symbol character tokenizer).
	An API mirror of ONE ~30-line reference function (transcribe.py:23-60): the order of its statements -- checkpoint args into `args`, frontend,
	text pipeline, model by class name, load_state_dict, eval / fuse / autocast, generator -- is the reference's, because a caller of the
	reference observes each of them (mutated args, loaded keys); the arithmetic behind every call is this package's."""
	torch.set_grad_enabled(False)
	checkpoint = args.checkpoint if isinstance(args.checkpoint, dict) else torch.load(args.checkpoint, map_location = 'cpu')
	args.sample_rate, args.window_size, args.window_stride, args.window, args.num_input_features = map(checkpoint['args'].get, ['sample_rate', 'window_size', 'window_stride', 'window', 'num_input_features'])
	frontend = models.LogFilterBankFrontend(
		args.num_input_features, args.sample_rate, args.window_size, args.window_stride, args.window,
		dither = getattr(args, 'dither', 0.0), dither0 = getattr(args, 'dither0', 0.0), normalize_signal = getattr(args, 'normalize_signal', True),
		debug_short_long_records_normalize_signal_multiplier = getattr(args, 'debug_short_long_records_normalize_signal_multiplier', 1.0))
	text_pipeline = getattr(args, 'text_pipeline', None) or TextPipeline(CharTokenizerLegacy(checkpoint['args'].get('alphabet', RU_ALPHABET)))
	model = getattr(models, getattr(args, 'model', None) or checkpoint['args']['model'])(
		args.num_input_features, [text_pipeline.tokenizer.vocab_size],
		frontend = frontend if getattr(args, 'frontend_in_model', True) else None,
		check_time_dim_padded = False,
		dict = lambda logits, log_probs, olen, **kwargs: (log_probs[0], logits[0], olen[0]),
		**checkpoint['args'].get('model_kwargs', {}))
	model.load_state_dict(checkpoint['model_state_dict'], strict = False)
	model = model.to(args.device)
	model.eval()
	model.fuse_conv_bn_eval()
	if str(args.device) != 'cpu':
		level = getattr(args, 'fp16', None)
		if level in models.JasperNet.SPLIT_DTYPES:  # (this package's own: args.fp16 = 'bf16x3' / 'f16x3' -- fp32-class logits from split-operand convs on the 16-bit matrix pipe)
			models.master_module(model).set_compute_dtype(level, inference = True)
		else:
			model, *_ = models.data_parallel_and_autocast(model, opt_level = level)
	generator = GreedyCTCGenerator()
	return text_pipeline, frontend, model, generator


def transcribe_batch(args, text_pipeline, model, generator, x, xlen, begin, end, y = None, ylen = None, segment_extra_info = None):
	"""One iteration of transcribe.main's loop (transcribe.py:140-200) on a collated batch: x (B, 1, T) or (B, T) waveform,
	xlen (B,) fractions, begin / end (B,) seconds, optional targets y (B, L, S) / ylen (B, L) for --align.
	Returns a namespace: log_probs, logits, olen, ts (per-frame time stamps), hyp_segments (per utterance: list of word segments with
	begin / end / hyp), hyp (joined strings) and, when args.align and targets are given, alignment (B, S), ref_segments."""
	device = torch.device(args.device)
	x = x.squeeze(1) if x.ndim == 3 else x
	log_probs, logits, olen = model(x.to(device), xlen.to(device))
	B = x.shape[0]
	duration = x.shape[-1] / args.sample_rate
	ts = duration * torch.linspace(0, 1, steps = log_probs.shape[-1], device = log_probs.device).unsqueeze(0).expand(B, -1)
	tokenizer = text_pipeline.tokenizer
	hyp_segments = [alternatives[0] for alternatives in generator.generate(tokenizer = tokenizer, log_probs = log_probs, begin = begin, end = end, output_lengths = olen, time_stamps = ts, segment_text_key = 'hyp', segment_extra_info = segment_extra_info)]
	hyp_segments = [map_text(text_pipeline.postprocess, hyp = hyp) for hyp in hyp_segments]
	out = types.SimpleNamespace(log_probs = log_probs, logits = logits, olen = olen, ts = ts, hyp_segments = hyp_segments, hyp = [join(hyp = h) for h in hyp_segments], alignment = None, ref_segments = None)
	if getattr(args, 'align', False) and y is not None and y.numel() > 0:
		y, ylen = y.to(device), ylen.to(device)
		out.alignment = ctc.alignment_bct(log_probs, y[:, 0, :], olen, ylen[:, 0], blank = tokenizer.eps_id)
		aligned_ts = ts.gather(1, out.alignment)
		one_hot = torch.nn.functional.one_hot(y[:, 0, :], num_classes = log_probs.shape[1]).permute(0, 2, 1).to(torch.float32)
		ref_segments = [alternatives[0] for alternatives in generator.generate(tokenizer = tokenizer, log_probs = one_hot, begin = begin, end = end, output_lengths = ylen[:, 0], time_stamps = aligned_ts, segment_text_key = 'ref', segment_extra_info = segment_extra_info)]
		out.ref_segments = [map_text(text_pipeline.postprocess, ref = ref) for ref in ref_segments]
	return out
