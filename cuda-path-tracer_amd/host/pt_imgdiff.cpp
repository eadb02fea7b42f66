// pt_imgdiff a b: compares two rendered images (PPM written by pt_headless, PNG or baseline JPEG):
// RMSE, largest difference, difference of the means against the Monte Carlo standard error of the
// mean (3-sigma test) and the share of identical values.  Exit code 0 when the images have the
// same size and the means agree within 3 sigma (or the images are identical), 1 otherwise.
#include "../../include/mipt_host.h"
#include <cstdio>

int main(int argc, char** argv)
{
	if (argc < 3) { fprintf(stderr, "usage: pt_imgdiff image_a image_b\n"); return 2; }
	float *a = nullptr, *b = nullptr; int32_t wa = 0, ha = 0, wb = 0, hb = 0;
	if (mipt_host_read_image_bgr255(argv[1], &a, &wa, &ha) != 0) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
	if (mipt_host_read_image_bgr255(argv[2], &b, &wb, &hb) != 0) { fprintf(stderr, "cannot read %s\n", argv[2]); mipt_host_free(a); return 2; }
	int rc = 1;
	if (wa != wb || ha != hb) {
		printf("sizes differ: %dx%d against %dx%d\n", wa, ha, wb, hb);
	} else {
		mipt_host_image_diff d;
		mipt_host_compare_images(a, b, (int64_t)3 * wa * ha, &d);
		bool ok = d.identical == 1.0 || d.mean_diff <= 3 * d.sigma;
		printf("%dx%d  rmse %.4f  max %.1f  |mean diff| %.4f  sigma %.4f  identical %.4f  -> %s\n", wa, ha, d.rmse, d.max_abs,
		       d.mean_diff, d.sigma, d.identical, ok ? "agree (3 sigma)" : "differ");
		rc = ok ? 0 : 1;
	}
	mipt_host_free(a); mipt_host_free(b);
	return rc;
}
