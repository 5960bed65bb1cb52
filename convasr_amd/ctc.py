"""Host mirror of the reference's ctc.py: forced alignment on the GPU."""
import torch

from . import ops


def alignment(log_probs, targets, input_lengths, target_lengths, blank: int = 0, pack_backpointers: bool = False):
	"""ctc.alignment (ctc.py:7-75): log_probs (T, B, C) like the reference, or this package's channels-last (B, C, T) tensor
	via `alignment_bct`.  Returns (B, S_max) int64: the last frame of the best path inside every label's state.
	pack_backpointers is accepted for signature parity; the kernel always packs (2 bits per state)."""
	lp = log_probs.float().permute(1, 0, 2).contiguous()
	return ops.ctc_alignment(lp, targets, input_lengths, target_lengths, blank)


def alignment_bct(log_probs_bct, targets, input_lengths, target_lengths, blank: int = 0):
	"""Same for the model's own output layout: logical (B, C, T) whose memory is (B, T, C)."""
	assert ops.is_cl(log_probs_bct)
	return ops.ctc_alignment(log_probs_bct.permute(0, 2, 1).float().contiguous(), targets, input_lengths, target_lengths, blank)
