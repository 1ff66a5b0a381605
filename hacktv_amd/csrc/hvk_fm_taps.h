/* hvk_fm_taps.h -- the FM video pre-emphasis filters (CCIR 405) of the reference, as the int16 values its
 * filter is initialised with.
 *
 * The reference does not design these filters, it carries them as literal tap tables (src/video.c:2017-2113: "test
 * taps" for 625 lines at 28 / 20.25 / 20 / 14 MHz and for 525 lines at 20.25 / 18 MHz) and picks one by line count
 * and sample rate (src/video.c:3690-3730; any other rate gets the 20.25 MHz table and a warning). There is nothing
 * to re-derive: the numbers ARE the specification. They are kept here already quantised the way fir_int16_init()
 * quantises them (lround(tap * 32767), src/fir.c:283), in the tables' own order, first tap first.
 * Generated once from the reference by the script in the comment at the end of this file. */
#ifndef HVK_FM_TAPS_H
#define HVK_FM_TAPS_H

#include <stdint.h>

typedef struct {
	int lines;              /* 625 or 525 */
	int sample_rate;        /* Hz; 0: the table for every other rate */
	int ntaps;
	const int16_t *q15;
} hvk_fm_taps_t;

/* src/video.c:2017 (67 taps) */
static const int16_t hvk_fm_625_28_taps[67] = {
	-1, -4, 0, 10, 14, -4, -32, -29, 24, 76, 44, -72,
	-146, -46, 168, 242, 9, -336, -357, 104, 601, 469, -351, -1003,
	-541, 842, 1641, 512, -1902, -2937, -213, 5674, 10904, 11328, 6078, -1514,
	-6616, -6776, -3473, -226, 647, -446, -1545, -1362, -277, 510, 386, -221,
	-537, -302, 115, 264, 88, -135, -167, -35, 79, 73, -2, -47,
	-31, 6, 19, 8, -4, -6, -1,
};

/* src/video.c:2031 (67 taps) */
static const int16_t hvk_fm_625_2025_taps[67] = {
	2, -3, -5, 8, 13, -16, -26, 28, 47, -45, -79, 68,
	125, -98, -190, 134, 279, -178, -401, 231, 567, -293, -794, 367,
	1115, -459, -1599, 584, 2425, -784, -4237, 1222, 12725, 15734, 4654, -7961,
	-9070, -2211, 1103, -1158, -2398, -315, 1030, -143, -996, -83, 627, 22,
	-488, -54, 333, 40, -236, -38, 157, 29, -102, -22, 62, 15,
	-36, -9, 19, 5, -8, -2, 3,
};

/* src/video.c:2045 (67 taps) */
static const int16_t hvk_fm_625_20_taps[67] = {
	2, -1, -8, 2, 17, -4, -33, 8, 58, -13, -95, 21,
	147, -32, -219, 47, 315, -67, -443, 94, 614, -129, -846, 178,
	1170, -247, -1656, 352, 2481, -537, -4289, 969, 12762, 15942, 4534, -8260,
	-9008, -1914, 1096, -1384, -2350, -72, 1017, -340, -975, 99, 618, -127,
	-480, 71, 330, -58, -235, 37, 157, -26, -103, 16, 64, -10,
	-37, 5, 19, -3, -9, 1, 3,
};

/* src/video.c:2059 (67 taps) */
static const int16_t hvk_fm_625_14_taps[67] = {
	-2, 2, 3, -11, 15, -3, -24, 45, -32, -25, 91, -103,
	18, 128, -221, 144, 103, -358, 383, -65, -437, 728, -464, -324,
	1117, -1197, 203, 1427, -2500, 1721, 1428, -6145, 10552, 22979, -1922, -14092,
	-616, -519, -3327, 1467, -428, -1125, 1200, -606, -293, 716, -566, 72,
	328, -400, 186, 88, -221, 167, -21, -90, 104, -45, -20, 46,
	-32, 4, 14, -13, 5, 2, -3,
};

/* src/video.c:2073 (71 taps) */
static const int16_t hvk_fm_525_2025_taps[71] = {
	2, 3, -6, -11, 8, 26, -4, -49, -15, 76, 58, -96,
	-130, 90, 231, -34, -347, -94, 449, 315, -494, -636, 421, 1048,
	-153, -1525, -418, 2031, 1506, -2540, -3828, 3107, 13778, 17016, 8308, -4179,
	-9294, -5526, -549, -181, -2419, -2699, -727, 454, -307, -1159, -682, 211,
	230, -331, -448, -39, 193, 1, -192, -101, 62, 55, -43, -59,
	-2, 24, 1, -16, -7, 2, 1, -1, -1, -1, -1,
};

/* src/video.c:2088 (67 taps) */
static const int16_t hvk_fm_525_18_taps[67] = {
	2, 0, -8, 1, 19, -3, -37, 5, 65, -9, -105, 15,
	162, -22, -241, 33, 346, -47, -487, 66, 675, -91, -930, 125,
	1286, -173, -1821, 247, 2731, -379, -4737, 691, 14206, 19200, 7608, -7232,
	-9777, -3015, 332, -2143, -3270, -819, 592, -683, -1365, -196, 483, -232,
	-623, -34, 297, -83, -285, 0, 151, -30, -121, 3, 63, -10,
	-42, 1, 19, -3, -10, 0, 2,
};

/* in the order src/video.c:3690-3730 tests them; the rate-0 entry of a line count comes last */
static const hvk_fm_taps_t hvk_fm_taps[] = {
	{ 625, 28000000, 67, hvk_fm_625_28_taps },
	{ 625, 20000000, 67, hvk_fm_625_20_taps },
	{ 625, 14000000, 67, hvk_fm_625_14_taps },
	{ 525, 18000000, 67, hvk_fm_525_18_taps },
	{ 625, 0, 67, hvk_fm_625_2025_taps },
	{ 525, 0, 71, hvk_fm_525_2025_taps },
	{ 0, 0, 0, 0 },
};

/* How this file was made (in the build container, where /root/reference exists): for each table named above,
 * the literals between the braces of `const static double <name>[]` in src/video.c were read as doubles and
 * printed as floor(|tap * 32767| + 0.5) with the tap's sign -- lround(). */

#endif
