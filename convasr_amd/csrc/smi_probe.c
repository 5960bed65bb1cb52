/* smi_probe.c -- host-side reader of one card's gpu_metrics table through librocm_smi64 (sysfs reads; no HIP / KFD compute call), for
 * bench.py's device-state sampler: the firmware's throttle residency accumulators say WHICH limiter holds the clock during the timed
 * region (PPT residency = d ppt_residency_acc / d accumulation_counter), which hwmon alone cannot.  Measurement infrastructure: not
 * part of libconvasr_hip.so and never on the product path.  Built with gcc by convasr_amd/build.py into libconvasr_smi.so. */
#include <stdint.h>
#include <string.h>
#include <rocm_smi/rocm_smi.h>

static int g_inited = 0;

/* bus / device / function of the card (what torch.cuda.get_device_properties reports); domain < 0: any.  Returns the rsmi device index or a negative code. */
int convasr_smi_open(int domain, int bus, int device) {
	if (!g_inited) {
		if (rsmi_init(0) != RSMI_STATUS_SUCCESS) return -1;
		g_inited = 1;
	}
	uint32_t n = 0;
	if (rsmi_num_monitor_devices(&n) != RSMI_STATUS_SUCCESS) return -2;
	for (uint32_t i = 0; i < n; ++i) {
		uint64_t bdf = 0;
		if (rsmi_dev_pci_id_get(i, &bdf) != RSMI_STATUS_SUCCESS) continue;
		const int d = (int)(bdf >> 32), b = (int)((bdf >> 8) & 0xff), dev = (int)((bdf >> 3) & 0x1f);
		if ((domain < 0 || d == domain) && b == bus && dev == device) return (int)i;
	}
	return n == 1 ? 0 : -3;
}

/* out[0] accumulation_counter, [1] ppt_residency_acc, [2] prochot, [3] socket thermal, [4] vr thermal, [5] hbm thermal, [6] socket power (W),
 * [7] mean of the valid current gfx clocks (MHz), [8] hotspot temperature (C), [9] hbm temperature (C).  Returns 0 or a negative code. */
int convasr_smi_sample(int dv, double* out) {
	rsmi_gpu_metrics_t m;
	memset(&m, 0, sizeof m);
	if (rsmi_dev_gpu_metrics_info_get((uint32_t)dv, &m) != RSMI_STATUS_SUCCESS) return -1;
	out[0] = (double)m.accumulation_counter;
	out[1] = (double)m.ppt_residency_acc;
	out[2] = (double)m.prochot_residency_acc;
	out[3] = (double)m.socket_thm_residency_acc;
	out[4] = (double)m.vr_thm_residency_acc;
	out[5] = (double)m.hbm_thm_residency_acc;
	out[6] = (double)m.current_socket_power;
	double s = 0; int k = 0;
	for (int i = 0; i < (int)(sizeof m.current_gfxclks / sizeof m.current_gfxclks[0]); ++i)
		if (m.current_gfxclks[i] != 0 && m.current_gfxclks[i] != 0xffff) { s += m.current_gfxclks[i]; ++k; }
	out[7] = k ? s / k : 0.0;
	out[8] = (double)m.temperature_hotspot;
	out[9] = (double)m.temperature_mem;
	return 0;
}
