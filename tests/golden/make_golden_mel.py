"""Mel filterbank fixtures from an implementation that is NOT this repository's: transformers.audio_utils.mel_filter_bank
(Hugging Face's restatement of librosa.filters.mel, norm='slaney', mel_scale='slaney', which its own test-suite checks against librosa).
librosa itself -- what /root/reference/models.py:522 calls -- is not installed in this image and not vendored by the reference, so this
is the closest independent pin available offline.  Writes tests/golden/mel_hf.npz (float64 matrices, [n_mels, nfft/2+1]).

	python tests/golden/make_golden_mel.py
"""
import os

import numpy as np
import transformers
from transformers.audio_utils import mel_filter_bank

CASES = [(16000, 512, 64), (8000, 256, 64), (16000, 512, 80), (16000, 400, 40), (22050, 1024, 128)]

if __name__ == '__main__':
	out = {'transformers_version': np.array(transformers.__version__)}
	for sr, nfft, n_mels in CASES:
		out[f'mel_{sr}_{nfft}_{n_mels}'] = mel_filter_bank(nfft // 2 + 1, n_mels, 0.0, sr / 2, sr, norm = 'slaney', mel_scale = 'slaney').T.astype(np.float64)
	np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'mel_hf.npz'), **out)
	print({k: getattr(v, 'shape', v) for k, v in out.items()})
