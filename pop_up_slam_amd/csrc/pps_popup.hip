// placeholder: replaced by the real pop-up kernels in the next commit
#include "../../include/pps.h"
extern "C" {
int pps_popup_planes(int, const float*, int, const float*, const float*, float*) { return PPS_ESTATE; }
int pps_popup_create(int, int, int, const float*, pps_popup**) { return PPS_ESTATE; }
int pps_popup_destroy(pps_popup*) { return PPS_ESTATE; }
int pps_popup_frame(pps_popup*, const float*, int, const float*, const float*, const int*, int, const unsigned char*, float, float,
                    float*, float*, unsigned char*, unsigned char*, float*, int*) { return PPS_ESTATE; }
int pps_popup_refresh_measurements(pps_graph*, int, const int*, const int*, const float*, const float*, const int*) { return PPS_ESTATE; }
}
