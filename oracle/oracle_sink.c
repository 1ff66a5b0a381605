/* oracle/oracle_sink.c -- TEST INFRASTRUCTURE (not product code).
 *
 * CPU restatement of the file sink's sample-format conversions
 * (src/rf_file.c:34-277): int16 I/Q pairs -> {u8, i8, u16, i16, i32, f32} x
 * {real (I only), complex}. Type codes are the reference's RF_UINT8 .. RF_FLOAT
 * (src/rf.h:31-36).
 */
#include <stdint.h>
#include <stddef.h>

long orc_sink_convert(const int16_t *iq, long samples, int type, int complex, void *dst)
{
	long n = complex ? samples * 2 : samples;
	long i;

	for(i = 0; i < n; i++)
	{
		int16_t v = complex ? iq[i] : iq[i * 2];

		switch(type)
		{
		case 0: ((uint8_t *)  dst)[i] = (v - INT16_MIN) >> 8; break;          /* :46, :166-167 */
		case 1: ((int8_t *)   dst)[i] = v >> 8; break;                        /* :68, :190-191 */
		case 2: ((uint16_t *) dst)[i] = (v - INT16_MIN); break;               /* :90, :214-215 */
		case 3: ((int16_t *)  dst)[i] = v; break;                             /* :112, :229 */
		case 4: ((int32_t *)  dst)[i] = (int32_t) ((uint32_t) v << 16) + v; break; /* :134, :247-248 */
		case 5: ((float *)    dst)[i] = (float) v * (1.0 / 32767.0); break;   /* :156, :271-272 */
		default: return(-1);
		}
	}

	return(n * (type <= 1 ? 1 : (type <= 3 ? 2 : 4)));
}

/* Two 64-bit sums over a block of I/Q pairs read as little-endian uint32 words w[0 .. n):
 *   s1 = sum w[i], s2 = sum (i + 1) * w[i], both modulo 2^64
 * -- a position-sensitive check cheap enough to be computed for every block of an hour of signal
 * (oracle/make_golden_hour.py over the reference's output; hvk_block_sums() over the device's). */
void orc_block_sums(const uint32_t *w, long n, uint64_t out[2])
{
	uint64_t s1 = 0, s2 = 0;
	long i;

	for(i = 0; i < n; i++)
	{
		s1 += w[i];
		s2 += (uint64_t) (i + 1) * w[i];
	}
	out[0] = s1;
	out[1] = s2;
}
