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


def bucket_ceiling(max_samples, sample_rate = 16000, window_stride = 0.01, time_padding_multiple = 128):
	"""The padded length (samples) every batch of one bucket can share: bucket k of train.py:597-601's bucket_fn holds the utterances with
	ceil((duration / window_stride + 1) / time_padding_multiple) = k, i.e. at most (k * time_padding_multiple - 1) hops of audio; that
	ceiling, rounded up to time_padding_multiple samples like collate_fn rounds (datasets.py:318).  Padding a batch to it instead of to its
	own longest utterance costs a few hundredths of a second of zeros per utterance (32 draws from a 1.28-s bucket nearly reach its top
	anyway) and leaves a training run with one batch shape per bucket -- what lets train.GraphedTrainStep replay a dozen captured graphs."""
	hop = int(round(sample_rate * window_stride))
	k = -(-(int(max_samples) + hop) // (hop * time_padding_multiple))
	top = (k * time_padding_multiple - 1) * hop
	return -(-top // time_padding_multiple) * time_padding_multiple


def _pad_on_gpu(tensors, rows, pad_multiple, device, pad_to = None):
	"""tensors: list of (rows, L_b) CPU tensors of one dtype -> ((B, rows, Lpad) device tensor, lengths list).
	pad_to (optional): callable longest length -> padded length (instead of rounding up to pad_multiple)."""
	lengths = [int(t.shape[-1]) for t in tensors]
	Lpad = int(math.ceil(max(lengths) / pad_multiple)) * pad_multiple if pad_to is None else int(pad_to(max(lengths)))
	assert Lpad >= max(lengths)
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


def collate_gpu(batch, time_padding_multiple = 128, device = None, speaker_missing = 0, pad_to = None):
	"""batch: list of (meta, speaker (S,), x (C, T), *targets (L,)) CPU samples, as AudioTextDataset.__getitem__ returns them in
	the default mode.  Returns (meta list, s, x, xlen, y, ylen) like collate_fn, with x / y / xlen / ylen on `device`.
	pad_to (optional): callable longest waveform length -> padded length of x (e.g. bucket_ceiling: one shape per bucket)."""
	device = torch.device('cuda', torch.cuda.current_device()) if device is None else device
	metas = [b[0] for b in batch]
	n_t = len(batch[0]) - 3
	x, xl, Tpad = _pad_on_gpu([b[2] for b in batch], len(batch[0][2]), time_padding_multiple, device, pad_to = pad_to)
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


class DistributedSamplerWrapper(torch.utils.data.Sampler):
	"""datasets.py:431-493 (a DistributedSampler with shuffle = False over the batches of another sampler): rank r of W takes
	batches r, r + W, r + 2 W, ... of what the wrapped sampler still has to deliver; the list is padded by wrapping around to a
	multiple of W, as torch's DistributedSampler does.  BucketingBatchSampler emits W consecutive batches per bucket group, so
	the W ranks of one iteration always get batches of the same bucket (equal padded length: balanced steps)."""

	def __init__(self, sampler, num_replicas = None, rank = None, shuffle = False, seed = 0):
		import torch.distributed as dist
		self.sampler = sampler
		self.num_replicas = num_replicas if num_replicas is not None else dist.get_world_size()
		self.rank = rank if rank is not None else dist.get_rank()
		self.epoch = 0
		self.shuffle, self.seed = shuffle, seed  # shuffle = True (never passed by the reference's train.py:643): torch.utils.data.DistributedSampler's permutation of the BATCHES, seeded seed + epoch

	def __iter__(self):
		batches = list(self.sampler)
		if self.shuffle:
			g = torch.Generator()
			g.manual_seed(self.seed + self.epoch)
			batches = [batches[i] for i in torch.randperm(len(batches), generator = g).tolist()]
		total = int(math.ceil(len(batches) / self.num_replicas)) * self.num_replicas
		if batches and total > len(batches):
			batches = (batches * int(math.ceil(total / len(batches))))[:total]
		return iter(batches[self.rank:total:self.num_replicas])

	def __len__(self):
		return int(math.ceil(len(self.sampler) / self.num_replicas))

	def state_dict(self):
		return self.sampler.state_dict()

	def load_state_dict(self, state_dict):
		self.sampler.load_state_dict(state_dict)

	def set_epoch(self, epoch):
		self.epoch = epoch
		self.sampler.set_epoch(epoch)

	@property
	def batch_idx(self):
		return self.sampler.batch_idx

	@batch_idx.setter
	def batch_idx(self, value):
		self.sampler.batch_idx = value


class SyntheticAudioTextDataset(torch.utils.data.Dataset):
	"""Stand-in for AudioTextDataset (datasets.py:23-355) with the same sample contract and `bucket` attribute, for benchmarks and
	tests: there is no audio decoding or text pipeline here (out of scope), utterance k is seeded noise of a fixed random duration
	and a random label string.  __getitem__ -> (meta, speaker (1,), x (1, T) float32, y (L,) int64), the tuple collate_fn /
	collate_gpu consume in the default mode; bucket[k] = ceil((duration / window_stride + 1) / time_padding_multiple), the
	bucket_fn of train.py:597-601."""

	def __init__(self, num_examples, min_duration = 5.0, max_duration = 20.0, sample_rate = 16000, window_stride = 0.01, time_padding_multiple = 128, num_labels = 37, labels_per_second = 5.0, seed = 0):
		g = torch.Generator().manual_seed(seed)
		self.sample_rate, self.num_labels, self.seed, self.window_stride = sample_rate, num_labels, seed, window_stride
		self.time_padding_multiple = time_padding_multiple
		self.duration = min_duration + (max_duration - min_duration) * torch.rand(num_examples, generator = g)
		self.num_samples = (self.duration * sample_rate).long()
		self.duration = self.num_samples.double() / sample_rate
		self.bucket = ((self.duration / window_stride + 1) / time_padding_multiple).ceil().to(torch.short)
		self.target_len = (self.duration * labels_per_second).long().clamp(min = 1)

	def __len__(self):
		return len(self.num_samples)

	def __getitem__(self, k):
		g = torch.Generator().manual_seed(self.seed * 1000003 + int(k) + 1)
		T, L = int(self.num_samples[k]), int(self.target_len[k])
		x = torch.rand(1, T, generator = g) * 2 - 1
		y = torch.randint(0, self.num_labels, (L, ), generator = g)
		meta = dict(example_id = int(k), duration = float(self.duration[k]), begin = 0.0, end = float(self.duration[k]))
		return meta, torch.zeros(1, dtype = torch.int64), x, y


def keep_samples(batch):
	"""DataLoader collate_fn for worker processes: the samples stay a list; padding happens on the GPU (collate_gpu)."""
	return batch


def gpu_batches(dataset, batch_sampler, device, num_workers = 0, time_padding_multiple = None, timeout = 0, pad_to_bucket = False):
	"""The reference's train DataLoader (train.py:647-655) with the batch assembly moved to the GPU: worker processes (or the main
	process) produce lists of ragged CPU samples, collate_gpu packs each list into one pinned buffer, copies it once and pads on
	the device.  Yields (meta, s, x, xlen, y, ylen) like the reference's loader, with x / xlen / y / ylen already on `device`;
	x is (B, T) for a one-row waveform dataset (what model(x.squeeze(1), ...) consumes, train.py:748).
	pad_to_bucket: pad every batch to its bucket's ceiling (bucket_ceiling) instead of to its own longest utterance: one batch shape per
	bucket, for train.GraphedTrainStep."""
	loader = torch.utils.data.DataLoader(dataset, batch_sampler = batch_sampler, collate_fn = keep_samples, num_workers = num_workers, pin_memory = False, timeout = timeout if num_workers > 0 else 0)
	mult = time_padding_multiple or getattr(dataset, 'time_padding_multiple', 128)
	pad_to = None
	if pad_to_bucket:
		sr, ws = getattr(dataset, 'sample_rate', 16000), getattr(dataset, 'window_stride', 0.01)
		pad_to = lambda longest: bucket_ceiling(longest, sr, ws, mult)
	for samples in loader:
		meta, s, x, xlen, y, ylen = collate_gpu(samples, time_padding_multiple = mult, device = device, pad_to = pad_to)
		yield meta, s, (x.squeeze(1) if x.shape[1] == 1 else x), xlen, y, ylen
