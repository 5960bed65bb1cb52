"""Host mirror of the reference's optimizers.py for the flat-arena training step.

NovoGrad (optimizers.py:66-90) is ONE fused launch pair over FlatParameters (per-tensor squared gradient norms, then the
update with torch.nn.utils.clip_grad_norm_ of train.py:777 folded in) instead of ~8 elementwise launches per parameter
tensor; it replaces both the reference's Python loop and apex's FusedNovoGrad (train.py:674).  The learning-rate schedules
are host arithmetic and keep the reference's class names and arguments."""
import torch

from . import ops, _lib
from . import functional as Fn


def _advance(opt, pair):
	"""The fused launch read pair[opt._cur] and wrote the other row: make that one current -- by swapping, or, once a step graph has been
	captured from this optimizer (its kernels read the row that was current at capture time at EVERY replay), by enqueueing the copy that
	hands the new values back.  From the first capture on (`_pinned`, set by train.GraphedTrainStep) the current row never moves again, in
	eager steps either: an eager step between two replays -- the warm-up of a new batch shape, a shape beyond max_graphs -- that flipped the
	rows would leave every existing graph reading the row that is one step old."""
	if Fn.capturing() or getattr(opt, '_pinned', False):
		_lib.call('convasr_copy', _lib.ptr(pair[1 - opt._cur]), _lib.ptr(pair[opt._cur]), pair[0].numel() * pair.element_size(), _lib.stream_ptr())
	else:
		opt._cur = 1 - opt._cur


class NovoGrad:
	clips_in_step = True  # the fused step computes the total gradient norm itself (total_norm) and applies clip_grad_norm_'s factor: train_step skips the separate sumsq pass

	def __init__(self, flat, lr = 1.0, betas = (0.95, 0.98), eps = 1e-8, weight_decay = 0.0, dampening = False):
		self.flat = flat
		self.defaults = dict(lr = lr, betas = betas, eps = eps, weight_decay = weight_decay, dampening = dampening)
		self.param_groups = [dict(params = flat.params, **self.defaults)]
		dev = flat.data.device
		n_seg = len(flat.params)
		host_offsets = list(flat.offsets) + [flat.numel]
		self.offsets = torch.tensor(host_offsets, dtype = torch.int64, device = dev)
		self._table = ops.novograd_work_table(host_offsets, dev)
		self.momentum_buffer = torch.zeros_like(flat.data)
		self.n_seg = n_seg
		self.grads_ema = torch.zeros(2, n_seg + 1, dtype = torch.float32, device = dev)  # [_cur] is current; last element = number of steps applied so far (a gated step does not count), kept on the device
		self._g2 = torch.zeros(n_seg, dtype = torch.float64, device = dev)
		self.total_norm = torch.zeros(1, dtype = torch.float32, device = dev)
		self.steps = 0
		self._cur = 0  # which row of grads_ema is current (the fused launch reads it and writes the other one)
		self.lr_dev = None  # see train.SGD

	def zero_grad(self, set_to_none = False):
		self.flat.zero_grad()

	def step(self, loss_gate = None):
		g = self.param_groups[0]
		flat = self.flat
		if flat.clip is None:
			flat.finalize_grads()
		max_norm = flat.clip[1] if flat.clip is not None else 0.0
		cur = self._cur
		grad_scale = flat.grad_scale
		scaler = getattr(flat, 'loss_scaler', None)
		flat.mirror_carried_over(lambda p16: ops.novograd_step(flat.data, flat.grad, self.momentum_buffer, self.grads_ema[cur], self.grads_ema[1 - cur], self._g2, self.offsets, flat.numel, self._table, max_norm, g['lr'], g['betas'][0], g['betas'][1], g['eps'], g['weight_decay'], g['dampening'], -1, loss_gate = loss_gate, total_norm = self.total_norm, grad_scale = grad_scale, p16 = p16, scaler = None if scaler is None else scaler.pair(), lr_dev = self.lr_dev if Fn.capturing() else None))
		if scaler is not None:
			scaler.advance()
		_advance(self, self.grads_ema)
		self.steps += 1
		flat.clip, flat.grad_scale = None, 1.0

	@property
	def state(self):
		"""Per-parameter view in the reference's vocabulary: {'_grads_ema': 0-d tensor, 'momentum_buffer': tensor}."""
		ema = self.grads_ema[self._cur, :self.n_seg]
		views = self.flat.param_views(self.momentum_buffer)  # the parameters' logical shapes: conv segments of the arena are tap-major
		return {p: dict(_grads_ema = ema[i], momentum_buffer = views[i]) for i, p in enumerate(self.flat.params)}

	def state_dict(self):
		"""format 2: momentum per parameter in the reference's shapes; grads_ema = one value per parameter, steps_applied = how many
		steps were not skipped by the device-side gates (what decides 'first step' on the device)."""
		ema = self.grads_ema[self._cur]
		return dict(format = 2, steps = self.steps, steps_applied = int(ema[self.n_seg].item()), momentum_buffer = self.flat.export_state(self.momentum_buffer), grads_ema = ema[:self.n_seg].clone(), param_groups = [{k: v for k, v in g.items() if k != 'params'} for g in self.param_groups])

	def load_state_dict(self, sd):
		self.steps = sd['steps']
		self.flat.import_state(self.momentum_buffer, sd['momentum_buffer'], 'NovoGrad.load_state_dict(momentum_buffer)')
		ema, cur = sd['grads_ema'], self.grads_ema[self._cur]
		if ema.numel() == self.n_seg + 1:  # (an earlier layout carried the applied-step counter behind the EMAs)
			cur.copy_(ema)
		elif ema.numel() == self.n_seg:
			cur[:self.n_seg].copy_(ema)
			cur[self.n_seg] = float(sd.get('steps_applied', sd['steps']))
		else:
			raise ValueError(f'NovoGrad.load_state_dict: grads_ema has {ema.numel()} entries for {self.n_seg} parameters')
		for g, s in zip(self.param_groups, sd['param_groups']):
			g.update(s)


class AdamW:
	"""torch.optim.AdamW (train.py:663-668: lr, betas, weight_decay from the command line; eps 1e-8, no amsgrad) as ONE fused launch over
	FlatParameters, with clip_grad_norm_, the device-side loss gate, the data-parallel mean and the fp16 loss scaler folded in like
	convasr_amd.train.SGD.  The count of APPLIED steps (what the bias corrections use) lives on the device: a gated / overflowed step
	does not advance it, as in a reference run that skipped optimizer.step()."""

	def __init__(self, flat, lr = 1e-3, betas = (0.9, 0.999), eps = 1e-8, weight_decay = 1e-2, amsgrad = False):
		if amsgrad:
			raise ValueError('AdamW(amsgrad = True) is not implemented (the reference never passes it)')
		self.flat = flat
		self.defaults = dict(lr = lr, betas = tuple(betas), eps = eps, weight_decay = weight_decay)
		self.param_groups = [dict(params = flat.params, **self.defaults)]
		self.exp_avg = torch.zeros_like(flat.data)
		self.exp_avg_sq = torch.zeros_like(flat.data)
		self.applied = torch.zeros(2, 1, dtype = torch.float32, device = flat.data.device)  # [_cur] is current
		self.steps = 0
		self._cur = 0
		self.lr_dev = None  # see train.SGD

	def zero_grad(self, set_to_none = False):
		self.flat.zero_grad()

	def step(self, loss_gate = None):
		g = self.param_groups[0]
		flat = self.flat
		if flat.clip is None:
			flat.finalize_grads()
		sumsq, max_norm = flat.clip if flat.clip is not None else (None, 0.0)
		scaler = getattr(flat, 'loss_scaler', None)
		if scaler is not None and sumsq is None:  # the overflow check reads the gradient's sum of squares
			sumsq = ops.sumsq(flat.grad, flat._sumsq)
		cur = self._cur
		grad_scale = flat.grad_scale
		flat.mirror_carried_over(lambda p16: ops.adamw_step(flat.data, flat.grad, self.exp_avg, self.exp_avg_sq, flat.numel, sumsq, max_norm, g['lr'], g['betas'][0], g['betas'][1], g['eps'], g['weight_decay'], self.applied[cur], self.applied[1 - cur], loss_gate = loss_gate, grad_scale = grad_scale, p16 = p16, scaler = None if scaler is None else scaler.pair(), lr_dev = self.lr_dev if Fn.capturing() else None))
		if scaler is not None:
			scaler.advance()
		_advance(self, self.applied)
		self.steps += 1
		flat.clip, flat.grad_scale = None, 1.0

	@property
	def state(self):
		"""torch's vocabulary: {param: dict(step, exp_avg, exp_avg_sq)} with the moments viewed in the parameters' logical shapes."""
		m, v = self.flat.param_views(self.exp_avg), self.flat.param_views(self.exp_avg_sq)
		step = self.applied[self._cur, 0]
		return {p: dict(step = step, exp_avg = m[i], exp_avg_sq = v[i]) for i, p in enumerate(self.flat.params)}

	def state_dict(self):
		"""format 2 (like SGD / NovoGrad here): moments per parameter in the reference's shapes, independent of the arena's element order."""
		return dict(format = 2, steps = self.steps, steps_applied = int(self.applied[self._cur, 0].item()), exp_avg = self.flat.export_state(self.exp_avg), exp_avg_sq = self.flat.export_state(self.exp_avg_sq),
			param_groups = [{k: v for k, v in g.items() if k != 'params'} for g in self.param_groups])

	def load_state_dict(self, sd):
		if 'state' in sd and 'steps' not in sd:
			# torch.optim.AdamW.state_dict() as the reference saves it (train.py:331) and reloads it (train.py:681-682): state[i] = dict(step,
			# exp_avg, exp_avg_sq) in parameter order, param_groups[0] carrying the hyper-parameters
			order = [i for g in sd['param_groups'] for i in g['params']]
			if len(order) != len(self.flat.params) or any(i not in sd['state'] for i in order):
				raise ValueError(f'AdamW.load_state_dict: a torch.optim.AdamW state for {len(order)} parameters ({len(sd["state"])} with moments) does not match the {len(self.flat.params)} trainable parameters of this arena')
			steps = {int(sd['state'][i]['step']) for i in order}
			if len(steps) != 1:
				raise ValueError(f'AdamW.load_state_dict: parameters with different step counts {sorted(steps)} (the fused kernel keeps one applied-step counter)')
			self.steps = steps.pop()
			self.flat.import_state(self.exp_avg, [sd['state'][i]['exp_avg'] for i in order], 'AdamW.load_state_dict(state[i].exp_avg)')
			self.flat.import_state(self.exp_avg_sq, [sd['state'][i]['exp_avg_sq'] for i in order], 'AdamW.load_state_dict(state[i].exp_avg_sq)')
			self.applied[self._cur, 0] = float(self.steps)
			for g, s in zip(self.param_groups, sd['param_groups']):
				g.update({k: (tuple(v) if k == 'betas' else v) for k, v in s.items() if k in ('lr', 'betas', 'eps', 'weight_decay')})
			return
		self.steps = sd['steps']
		self.flat.import_state(self.exp_avg, sd['exp_avg'], 'AdamW.load_state_dict(exp_avg)')
		self.flat.import_state(self.exp_avg_sq, sd['exp_avg_sq'], 'AdamW.load_state_dict(exp_avg_sq)')
		self.applied[self._cur, 0] = float(sd.get('steps_applied', sd['steps']))
		for g, s in zip(self.param_groups, sd['param_groups']):
			g.update(s)


def reset_options(optimizer):
	"""optimizers.py:4-6: every parameter group back to the optimizer's constructor defaults."""
	for group in optimizer.param_groups:
		group.update(optimizer.defaults)


class LRScheduler:
	"""optimizers.py:9-16.  `step(n)` writes the schedule's rates for iteration n into the optimizer's parameter groups; the reference calls it
	with the iteration number after every applied optimizer step (train.py:783).  Host arithmetic only: a captured step graph reads the rate from
	device memory, which train.GraphedTrainStep refreshes whenever `param_groups[0]['lr']` moved."""

	def __init__(self, optimizer):
		self.optimizer = optimizer

	def initial_rates(self):
		return [float(group['lr']) for group in self.optimizer.param_groups]

	def get_lr(self, step):
		raise NotImplementedError

	def step(self, step):
		rates = self.get_lr(step)
		for i, group in enumerate(self.optimizer.param_groups):
			group['lr'] = rates[i]


class NoopLR(LRScheduler):
	"""optimizers.py:18-20: the rates stay what they are."""

	def get_lr(self, step):
		return self.initial_rates()


class MultiStepLR(LRScheduler):
	"""optimizers.py:23-32: rate x gamma^(index + 1 of the LAST milestone in list order that the step has reached) -- the reference's own reading of
	its milestone list, kept as it is (for an ascending list: the number of milestones passed)."""

	def __init__(self, optimizer, gamma, milestones):
		super().__init__(optimizer)
		self.init_lr, self.gamma, self.milestones = self.initial_rates(), gamma, milestones

	def get_lr(self, step):
		power = 0
		for i, m in enumerate(self.milestones):
			if step >= m:
				power = i + 1
		return [lr0 * self.gamma ** power for lr0 in self.init_lr]


class PolynomialDecayLR(LRScheduler):
	"""Linear warm-up to the initial rate, then polynomial decay to end_lr over decay_steps (optimizers.py:36-63; the
	reference's decay branch reads an undefined name and cannot run, this is the schedule its arguments describe)."""

	def __init__(self, optimizer, decay_steps, power = 1.0, begin_decay_at = 0, end_lr = 0.0, warmup_steps = 0):
		super().__init__(optimizer)
		self.decay_steps, self.power, self.begin_decay_at, self.end_lr, self.warmup_steps = decay_steps, power, begin_decay_at, end_lr, warmup_steps
		self.init_lr = self.initial_rates()

	def _rate(self, lr0, step):
		if step >= self.begin_decay_at:  # (the decay branch wins over the warm-up one where both apply, as in the reference's statement order)
			k = min(step - self.begin_decay_at, self.decay_steps)
			if k >= self.decay_steps:
				return self.end_lr
			return self.end_lr + (lr0 - self.end_lr) * ((self.decay_steps - k) / self.decay_steps) ** self.power
		if self.warmup_steps > 0 and step < self.warmup_steps:
			return lr0 * step / self.warmup_steps
		return lr0

	def get_lr(self, step):
		return [self._rate(lr0, step) for lr0 in self.init_lr]
