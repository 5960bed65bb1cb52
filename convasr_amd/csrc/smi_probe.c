/* smi_probe.c -- host-side reader of one card's gpu_metrics table through librocm_smi64 (sysfs reads; no HIP / KFD compute call), for
 * bench.py's device-state sampler: the firmware's throttle residency accumulators say WHICH limiter holds the clock during the timed
 * region (PPT residency = d ppt_residency_acc / d accumulation_counter), which hwmon alone cannot.  Measurement infrastructure: not
 * part of libconvasr_hip.so and never on the product path.  Built with gcc by convasr_amd/build.py into libconvasr_smi.so. */
#include <fcntl.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
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

/* ---- hwmon sampler: a native thread (no Python in the sampling loop: a Python sampler thread takes the GIL ~200 times a second from the
 * thread that is launching kernels, which measured as +1 % on the eager Wav2Letter step) reads power1_input (microwatts) and freq1_input
 * (Hz) of one hwmon directory every interval_us and keeps running sums. */
static pthread_t g_thread;
static volatile int g_run = 0;
static char g_power_path[512], g_freq_path[512];
static int g_interval_us = 5000;
static double g_sum_power = 0, g_sum_freq = 0;
static long g_n_power = 0, g_n_freq = 0, g_n = 0;

static int read_ll(const char* path, long long* v) {
	char buf[64];
	const int fd = open(path, O_RDONLY);
	if (fd < 0) return -1;
	const ssize_t n = read(fd, buf, sizeof buf - 1);
	close(fd);
	if (n <= 0) return -1;
	buf[n] = 0;
	*v = atoll(buf);
	return 0;
}

static void* sampler(void* arg) {
	(void)arg;
	struct timespec ts = {0, 0};
	ts.tv_nsec = (long)g_interval_us * 1000L;
	while (g_run) {
		long long v;
		if (read_ll(g_power_path, &v) == 0 && v > 0) { g_sum_power += (double)v; ++g_n_power; }
		if (read_ll(g_freq_path, &v) == 0 && v > 0) { g_sum_freq += (double)v; ++g_n_freq; }
		++g_n;
		nanosleep(&ts, NULL);
	}
	return NULL;
}

int convasr_hwmon_start(const char* hwmon_dir, int interval_us) {
	if (g_run || !hwmon_dir) return -1;
	snprintf(g_power_path, sizeof g_power_path, "%s/power1_input", hwmon_dir);
	snprintf(g_freq_path, sizeof g_freq_path, "%s/freq1_input", hwmon_dir);
	g_interval_us = interval_us > 0 ? interval_us : 5000;
	g_sum_power = g_sum_freq = 0; g_n_power = g_n_freq = g_n = 0;
	g_run = 1;
	if (pthread_create(&g_thread, NULL, sampler, NULL) != 0) { g_run = 0; return -2; }
	return 0;
}

/* out[0] mean power (W), out[1] mean shader clock (MHz), out[2] samples taken; returns 0 */
int convasr_hwmon_stop(double* out) {
	if (!g_run) return -1;
	g_run = 0;
	pthread_join(g_thread, NULL);
	out[0] = g_n_power ? g_sum_power / g_n_power / 1e6 : 0.0;
	out[1] = g_n_freq ? g_sum_freq / g_n_freq / 1e6 : 0.0;
	out[2] = (double)g_n;
	return 0;
}
