"""Host mirror of the two pieces of the reference's datasets.py that feed the hot path with variable-length batches (SURVEY 8(f)
row f4): the bucketed batch schedule and the batch assembly.  File reading, feature extraction on the CPU and the text pipeline
stay out of scope.

* BucketingBatchSampler (datasets.py:357-401): same constructor, set_epoch / state_dict protocol and -- given the same epoch --
  the same batches as the reference (it draws from one torch.Generator in the same order).
* collate_gpu: AudioTextDataset.collate_fn (datasets.py:305-332) with the padding done on the GPU: the ragged samples of a batch
  travel as ONE pinned, packed host buffer (one copy instead of B), a kernel scatters them into the zero-padded (B, C, Tpad) batch."""
import math

import torch

from . import _lib


class BucketingBatchSampler(torch.utils.data.Sampler):
	def __init__(self, dataset, batch_size = 1, world_size = 1):
		self.dataset, self.batch_size, self.world_size = dataset, batch_size, world_size  # world_size consecutive batches come from one bucket
		self.buckets = {int(k): (dataset.bucket == k).nonzero(as_tuple = True)[0] for k in dataset.bucket.unique()}
		self.batch_idx = 0
		self.set_epoch(0)

	def set_epoch(self, epoch):
		rng = torch.Generator()
		rng.manual_seed(epoch)
		unit = self.batch_size * self.world_size
		parts = []
		for members in self.buckets.values():
			need = int(math.ceil(len(members) / unit)) * unit
			repeats = torch.randint(0, len(members), size = (need - len(members), ), generator = rng, device = members.device)
			padded = torch.cat([members, members[repeats]])
			parts.append(padded[torch.randperm(len(padded), generator = rng)].reshape(-1, self.batch_size))
		batches = torch.cat(parts)
		assert len(batches) % self.world_size == 0
		groups = torch.randperm(len(batches) // self.world_size, generator = rng)
		self.shuffled = batches[torch.arange(len(batches)).view(-1, self.world_size)[groups].flatten()]

	def __iter__(self):
		return iter(self.shuffled[self.batch_idx:])

	def __len__(self):
		return len(self.shuffled)

	def state_dict(self):
		return dict(batch_idx = self.batch_idx)

	def load_state_dict(self, state_dict):
		self.batch_idx = state_dict['batch_idx']


def _pad_on_gpu(tensors, rows, pad_multiple, device):
	"""tensors: list of (rows, L_b) CPU tensors of one dtype -> ((B, rows, Lpad) device tensor, lengths list)."""
	lengths = [int(t.shape[-1]) for t in tensors]
	Lpad = int(math.ceil(max(lengths) / pad_multiple)) * pad_multiple
	dtype = tensors[0].dtype
	packed = torch.empty(sum(l * rows for l in lengths), dtype = dtype).pin_memory()
	offsets, pos = [], 0
	for t, l in zip(tensors, lengths):
		packed[pos:pos + l * rows].copy_(t.reshape(-1))
		offsets.append(pos)
		pos += l * rows
	meta = torch.tensor([offsets, lengths], dtype = torch.int64).pin_memory().to(device, non_blocking = True)
	dev_packed = packed.to(device, non_blocking = True)
	out = torch.empty(len(tensors), rows, Lpad, dtype = dtype, device = device)
	_lib.call('convasr_collate_pad', _lib.ptr(dev_packed), _lib.ptr(meta[0]), _lib.ptr(meta[1]), _lib.ptr(out), dtype.itemsize, len(tensors), rows, Lpad, _lib.stream_ptr())
	return out, lengths, Lpad


def collate_gpu(batch, time_padding_multiple = 128, device = None, speaker_missing = 0):
	"""batch: list of (meta, speaker (S,), x (C, T), *targets (L,)) CPU samples, as AudioTextDataset.__getitem__ returns them in
	the default mode.  Returns (meta list, s, x, xlen, y, ylen) like collate_fn, with x / y / xlen / ylen on `device`."""
	device = torch.device('cuda', torch.cuda.current_device()) if device is None else device
	metas = [b[0] for b in batch]
	n_t = len(batch[0]) - 3
	x, xl, Tpad = _pad_on_gpu([b[2] for b in batch], len(batch[0][2]), time_padding_multiple, device)
	xlen = torch.tensor([l / Tpad if Tpad > 0 else 1.0 for l in xl], dtype = torch.float32).to(device, non_blocking = True)
	ys, yl = [], []
	Lpad = max(int(math.ceil(max(b[3 + j].shape[-1] for b in batch) / time_padding_multiple)) * time_padding_multiple for j in range(n_t)) if n_t else 0
	y = torch.zeros(len(batch), n_t, Lpad, dtype = torch.int64, device = device)
	for j in range(n_t):
		yj, lj, _ = _pad_on_gpu([b[3 + j].reshape(1, -1).to(torch.int64) for b in batch], 1, time_padding_multiple, device)
		y[:, j, :yj.shape[-1]] = yj[:, 0]
		yl.append(lj)
	ylen = torch.tensor(yl, dtype = torch.int64).t().contiguous().to(device, non_blocking = True) if n_t else torch.zeros(len(batch), 0, dtype = torch.int64, device = device)
	Smax = max(b[1].shape[-1] for b in batch)
	s = torch.full((len(batch), Smax), speaker_missing, dtype = torch.int64)
	for k, b in enumerate(batch):
		s[k, :b[1].shape[-1]] = b[1]
	return metas, s, x, xlen, y, ylen
