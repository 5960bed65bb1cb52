"""Greedy CTC decoding (reference: transcript_generators.py:8-93, text_tokenizers.py:7-51).

The per-frame argmax over classes runs on the GPU (convasr_argmax); the collapse rules -- skip leading blank/space, merge
repeats unless a blank intervened, >= blank_amount_to_space consecutive blanks insert one space, a blank right after a space
is ignored, a new segment starts at every word-start token when time stamps are given -- stay a host loop over B x t ints,
as in the reference."""
import torch

from . import ops


class Segment(dict):
	pass


class Transcript(list):
	pass


class CharTokenizerLegacy:
	"""text_tokenizers.py:7-51: alphabet + ['*', '.', '2', ' ', '|']; eps ('|') is the CTC blank and the last class."""

	def __init__(self, alphabet):
		self.alphabet = alphabet
		self.idx2char = list(alphabet) + ['*', '.', '2', ' ', '|']
		self.char2idx = {c: i for i, c in enumerate(self.idx2char)}
		self.unk_idx, self.space_id, self.eps_id = self.char2idx['*'], self.char2idx[' '], self.char2idx['|']

	vocab = property(lambda self: self.idx2char)
	vocab_size = property(lambda self: len(self.idx2char))
	silence_tokens_ids = property(lambda self: {self.eps_id, self.space_id})

	def is_start_word_token(self, idx):
		return idx == self.space_id

	def encode(self, sentences, **kwargs):
		return [[self.char2idx.get(c, self.unk_idx) for c in s] for s in sentences]

	def decode(self, tokens, **kwargs):
		return [''.join(self.idx2char[i] for i in t) for t in tokens]


class GreedyCTCGenerator:
	def __init__(self, blank_amount_to_space = 10):
		self.blank_amount_to_space = blank_amount_to_space

	def generate(self, tokenizer, log_probs, begin, end, output_lengths = None, time_stamps = None, segment_text_key = 'hyp', segment_extra_info = None):
		idx_all = (ops.argmax(log_probs) if log_probs.is_cuda else log_probs.argmax(dim = 1)).cpu().tolist()
		ts_all = time_stamps.cpu().tolist() if time_stamps is not None else None
		begin = torch.clamp(begin, min = 0.0).cpu().tolist() if time_stamps is not None else begin.cpu().tolist()
		end = end.cpu().tolist()
		lens = output_lengths.cpu().tolist() if torch.is_tensor(output_lengths) else output_lengths
		silence, eps, space = tokenizer.silence_tokens_ids, tokenizer.eps_id, getattr(tokenizer, 'space_id', None)
		result = []
		for i, path in enumerate(idx_all):
			n = lens[i] if lens is not None else len(path)
			ts = ts_all[i] if ts_all is not None else None
			transcript = Transcript()
			start = next((t for t, c in enumerate(path) if c not in silence), len(path))
			if start >= len(path):
				result.append([transcript])
				continue
			tokens = [eps]
			t_begin = begin[i] + ts[start] if ts is not None else begin[i]
			t_end = end[i]
			blanks, repeat_ok = 0, False

			def flush():
				seg = Segment(begin = t_begin, end = t_end, **{segment_text_key: tokenizer.decode([tokens[1:]])[0]})
				if segment_extra_info is not None:
					seg.update(segment_extra_info[i])
				transcript.append(seg)

			for t in range(start, n):
				c = path[t]
				if c == eps:
					if tokens[-1] == space:
						continue
					repeat_ok = True
					blanks += 1
					if blanks >= self.blank_amount_to_space and not tokenizer.is_start_word_token(tokens[-1]):
						tokens.append(space)
					continue
				if c == tokens[-1] and not repeat_ok:
					continue
				if ts is not None and tokenizer.is_start_word_token(c):
					flush()
					tokens = [eps, c]
					t_begin = begin[i] + ts[t]
				repeat_ok = False
				tokens.append(c)
				t_end = begin[i] + ts[t] if ts is not None else end[i]
				blanks = 0
			if len(tokens) > 1:
				flush()
			result.append([transcript])
		return result
