// Baseline JPEG decoder for the harness (ITU-T T.81, sequential DCT, Huffman, 8 bits per sample,
// one or three components, sampling factors 1..4, restart intervals).  The reference loads its
// diffuse and normal maps through CImage (MainFrm.cpp:258-300), which takes JPEG files as well as
// PNG; the default Sponza scene names JPEG textures.  Progressive and arithmetic-coded files are
// refused.  Output is the DWORD layout of the PNG loader: B | G << 8 | R << 16 | 255 << 24.
//
// CImage decodes through GDI+, whose inverse DCT and chroma upsampling are not documented; this
// decoder uses the accurate integer inverse DCT of the IJG code base's description (Loeffler,
// Ligtenberg, Moschytz, 13-bit constants) and replicates chroma samples, so texels can differ from
// the reference's by a few levels out of 255.
#include "Scene.h"

#include <cstdio>
#include <cstring>
#include <vector>

namespace pth {
namespace {

struct Huff {
	// canonical code tables: for each length the first code, the first value index and the count
	int32_t mincode[17] = {}, maxcode[18] = {}, valptr[17] = {};
	uint8_t vals[256] = {};
	uint8_t look[512] = {};  // 9-bit prefix -> length (0 = longer than 9 bits)
	uint8_t lookVal[512] = {};
	bool present = false;
};

struct Component {
	int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
	int blocksW = 0, blocksH = 0;     // blocks per line / column, padded to whole MCUs
	int pred = 0;
	std::vector<uint8_t> plane;      // blocksW * 8 by blocksH * 8 samples
};

struct Reader {
	const uint8_t* p; const uint8_t* end;
	uint32_t bits = 0; int count = 0;
	bool marker = false;             // a marker was met inside the entropy-coded data
	int byte()
	{
		if (marker || p >= end) return 0;
		int b = *p++;
		if (b == 0xFF) {
			if (p < end && *p == 0) { p++; }            // stuffed zero
			else { p--; marker = true; return 0; }      // a real marker: feed zeros from here on
		}
		return b;
	}
	void fill() { while (count <= 24) { bits |= (uint32_t)byte() << (24 - count); count += 8; } }
	int peek(int n) { fill(); return (int)(bits >> (32 - n)); }
	void skip(int n) { bits <<= n; count -= n; }
	int get(int n) { if (n == 0) return 0; int v = peek(n); skip(n); return v; }
	void reset() { bits = 0; count = 0; marker = false; }
};

bool build_huff(Huff& H, const uint8_t* counts, const uint8_t* vals, int nvals)
{
	int code = 0, k = 0;
	memcpy(H.vals, vals, (size_t)nvals);
	memset(H.look, 0, sizeof(H.look));
	for (int len = 1; len <= 16; len++) {
		H.valptr[len] = k;
		H.mincode[len] = code;
		for (int i = 0; i < counts[len - 1]; i++, k++, code++) {
			if (len <= 9) {
				int first = code << (9 - len), n = 1 << (9 - len);
				if (first + n > 512) return false;
				for (int j = 0; j < n; j++) { H.look[first + j] = (uint8_t)len; H.lookVal[first + j] = vals[k]; }
			}
		}
		H.maxcode[len] = counts[len - 1] ? code - 1 : -1;
		if (code > (1 << len)) return false;
		code <<= 1;
	}
	H.maxcode[17] = 0x7fffffff;
	H.present = k == nvals;
	return H.present;
}

int decode_symbol(Reader& r, const Huff& H)
{
	int look = r.peek(9);
	int len = H.look[look];
	if (len) { r.skip(len); return H.lookVal[look]; }
	int code = r.peek(16);
	for (len = 10; len <= 16; len++) {
		int c = code >> (16 - len);
		if (H.maxcode[len] >= 0 && c <= H.maxcode[len] && c >= H.mincode[len]) {
			r.skip(len);
			return H.vals[H.valptr[len] + c - H.mincode[len]];
		}
	}
	r.skip(16);
	return -1;
}

inline int extend(int v, int n) { return (n && v < (1 << (n - 1))) ? v - (1 << n) + 1 : v; }

const uint8_t kZigzag[64] = {
	0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
	35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };

// Accurate integer inverse DCT (two passes, 13-bit constants, 2 extra bits between the passes).
const int CB = 13, PASS1 = 2;
inline int32_t fix(double x) { return (int32_t)(x * (1 << CB) + 0.5); }
inline int32_t descale(int64_t x, int n) { return (int32_t)((x + ((int64_t)1 << (n - 1))) >> n); }

void idct_1d(const int32_t* in, int stride, int32_t* out, int ostride, int shift, bool firstPass)
{
	static const int32_t F0_298 = fix(0.298631336), F0_390 = fix(0.390180644), F0_541 = fix(0.541196100), F0_765 = fix(0.765366865),
		F0_899 = fix(0.899976223), F1_175 = fix(1.175875602), F1_501 = fix(1.501321110), F1_847 = fix(1.847759065),
		F1_961 = fix(1.961570560), F2_053 = fix(2.053119869), F2_562 = fix(2.562915447), F3_072 = fix(3.072711026);
	int64_t z2 = in[2 * stride], z3 = in[6 * stride];
	int64_t z1 = (z2 + z3) * F0_541;
	int64_t tmp2 = z1 - z3 * F1_847, tmp3 = z1 + z2 * F0_765;
	int64_t tmp0 = ((int64_t)in[0] + in[4 * stride]) * (1 << CB), tmp1 = ((int64_t)in[0] - in[4 * stride]) * (1 << CB);
	int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
	int64_t t0 = in[7 * stride], t1 = in[5 * stride], t2 = in[3 * stride], t3 = in[1 * stride];
	int64_t y1 = t0 + t3, y2 = t1 + t2, y3 = t0 + t2, y4 = t1 + t3, y5 = (y3 + y4) * F1_175;
	t0 *= F0_298; t1 *= F2_053; t2 *= F3_072; t3 *= F1_501;
	y1 *= -F0_899; y2 *= -F2_562; y3 *= -F1_961; y4 *= -F0_390;
	y3 += y5; y4 += y5;
	t0 += y1 + y3; t1 += y2 + y4; t2 += y2 + y3; t3 += y1 + y4;
	(void)firstPass;
	out[0 * ostride] = descale(tmp10 + t3, shift); out[7 * ostride] = descale(tmp10 - t3, shift);
	out[1 * ostride] = descale(tmp11 + t2, shift); out[6 * ostride] = descale(tmp11 - t2, shift);
	out[2 * ostride] = descale(tmp12 + t1, shift); out[5 * ostride] = descale(tmp12 - t1, shift);
	out[3 * ostride] = descale(tmp13 + t0, shift); out[4 * ostride] = descale(tmp13 - t0, shift);
}

void idct_block(const int32_t* coef, uint8_t* dst, int pitch)
{
	int32_t ws[64], row[8];
	for (int c = 0; c < 8; c++) idct_1d(coef + c, 8, ws + c, 8, CB - PASS1, true);          // columns
	for (int r = 0; r < 8; r++) {
		idct_1d(ws + 8 * r, 1, row, 1, CB + PASS1 + 3, false);                              // rows, /8 and back to samples
		for (int c = 0; c < 8; c++) {
			int v = row[c] + 128;
			dst[r * pitch + c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
		}
	}
}

inline uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

} // namespace

static bool LoadJpegUnguarded(const char* path, std::vector<uint32_t>& out, int32_t& width, int32_t& height)
{
	FILE* f = fopen(path, "rb");
	if (!f) return false;
	std::vector<uint8_t> buf;
	uint8_t tmp[65536];
	size_t got;
	while ((got = fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + got);
	fclose(f);
	if (buf.size() < 4 || buf[0] != 0xFF || buf[1] != 0xD8) return false;

	uint16_t quant[4][64] = {};
	bool haveQuant[4] = {};
	Huff dc[4], ac[4];
	std::vector<Component> comps;
	int W = 0, H = 0, restartInterval = 0, hmax = 1, vmax = 1;
	size_t pos = 2;
	const size_t n = buf.size();
	auto u16 = [&](size_t at) { return (int)((buf[at] << 8) | buf[at + 1]); };

	while (pos + 4 <= n) {
		if (buf[pos] != 0xFF) { pos++; continue; }
		int m = buf[pos + 1];
		if (m == 0xFF) { pos++; continue; }
		pos += 2;
		if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
		if (m == 0xD9) break;
		if (pos + 2 > n) return false;
		int len = u16(pos);
		if (len < 2 || pos + (size_t)len > n) return false;
		const size_t seg = pos + 2, segEnd = pos + (size_t)len;
		if (m == 0xDB) {                                   // quantisation tables
			size_t q = seg;
			while (q < segEnd) {
				int pq = buf[q] >> 4, tq = buf[q] & 15; q++;
				if (tq > 3 || q + (size_t)(pq ? 128 : 64) > segEnd) return false;
				for (int i = 0; i < 64; i++) { quant[tq][kZigzag[i]] = (uint16_t)(pq ? u16(q + 2 * (size_t)i) : buf[q + (size_t)i]); }
				q += pq ? 128 : 64;
				haveQuant[tq] = true;
			}
		} else if (m == 0xC4) {                            // Huffman tables
			size_t q = seg;
			while (q + 17 <= segEnd) {
				int tc = buf[q] >> 4, th = buf[q] & 15;
				if (tc > 1 || th > 3) return false;
				const uint8_t* counts = &buf[q + 1];
				int total = 0;
				for (int i = 0; i < 16; i++) total += counts[i];
				if (total > 256 || q + 17 + (size_t)total > segEnd) return false;
				if (!build_huff(tc ? ac[th] : dc[th], counts, &buf[q + 17], total)) return false;
				q += 17 + (size_t)total;
			}
		} else if (m == 0xC0 || m == 0xC1) {               // sequential DCT, Huffman
			if (len < 8 || buf[seg] != 8) return false;
			H = u16(seg + 1); W = u16(seg + 3);
			int nc = buf[seg + 5];
			if (W <= 0 || H <= 0 || (nc != 1 && nc != 3) || seg + 6 + 3 * (size_t)nc > segEnd) return false;
			if (!comps.empty()) return false;                // a second frame header: not a baseline single-frame file
			if ((int64_t)W * H > ((int64_t)1 << 28)) return false;
			comps.assign((size_t)nc, Component());
			hmax = vmax = 1;
			for (int i = 0; i < nc; i++) {
				Component& c = comps[(size_t)i];
				c.id = buf[seg + 6 + 3 * (size_t)i]; c.h = buf[seg + 7 + 3 * (size_t)i] >> 4; c.v = buf[seg + 7 + 3 * (size_t)i] & 15;
				c.tq = buf[seg + 8 + 3 * (size_t)i];
				if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) return false;
				for (int k = 0; k < i; k++) if (comps[(size_t)k].id == c.id) return false;      // every component needs an id of its own
				if (c.h > hmax) hmax = c.h;
				if (c.v > vmax) vmax = c.v;
			}
		} else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
			return false;                                    // progressive, lossless, arithmetic: not supported
		} else if (m == 0xDD) {
			if (len < 4) return false;
			restartInterval = u16(seg);
		} else if (m == 0xDA) {                            // start of scan: the entropy-coded data follow
			if (comps.empty()) return false;
			int ns = buf[seg];
			if (ns != (int)comps.size() || seg + 1 + 2 * (size_t)ns + 3 > segEnd) return false;   // interleaved single scan only
			std::vector<bool> assigned(comps.size(), false);
			for (int i = 0; i < ns; i++) {
				int cid = buf[seg + 1 + 2 * (size_t)i], tabs = buf[seg + 2 + 2 * (size_t)i];
				Component* c = nullptr;
				for (size_t k = 0; k < comps.size(); k++) if (comps[k].id == cid) { c = &comps[k]; if (assigned[k]) return false; assigned[k] = true; }
				if (!c) return false;
				c->td = tabs >> 4; c->ta = tabs & 15;
				if (c->td > 3 || c->ta > 3 || !dc[c->td].present || !ac[c->ta].present || !haveQuant[c->tq]) return false;
			}
			const int mcuW = 8 * hmax, mcuH = 8 * vmax;
			const int mcusX = (W + mcuW - 1) / mcuW, mcusY = (H + mcuH - 1) / mcuH;
			if ((int64_t)mcusX * mcusY > (int64_t)1 << 24) return false;
			for (auto& c : comps) {
				c.blocksW = mcusX * c.h; c.blocksH = mcusY * c.v; c.pred = 0;
				c.plane.assign((size_t)c.blocksW * 8 * (size_t)c.blocksH * 8, 128);
			}
			Reader r{ &buf[segEnd], &buf[0] + n };
			int untilRestart = restartInterval;
			int32_t coef[64];
			for (int my = 0; my < mcusY; my++) {
				for (int mx = 0; mx < mcusX; mx++) {
					if (restartInterval && untilRestart == 0) {
						// skip to the restart marker, then start afresh
						const uint8_t* q = r.p;
						while (q + 1 < r.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) q++;
						if (q + 1 >= r.end) return false;
						r.p = q + 2; r.reset();
						for (auto& c : comps) c.pred = 0;
						untilRestart = restartInterval;
					}
					for (auto& c : comps) {
						for (int by = 0; by < c.v; by++) {
							for (int bx = 0; bx < c.h; bx++) {
								memset(coef, 0, sizeof(coef));
								int s = decode_symbol(r, dc[c.td]);
								if (s < 0 || s > 11) return false;
								c.pred += extend(r.get(s), s);
								coef[0] = c.pred * quant[c.tq][0];
								for (int k = 1; k < 64;) {
									int rs = decode_symbol(r, ac[c.ta]);
									if (rs < 0) return false;
									int run = rs >> 4, size = rs & 15;
									if (size == 0) { if (run == 15) { k += 16; continue; } break; }
									k += run;
									if (k > 63) return false;
									int zz = kZigzag[k];
									coef[zz] = extend(r.get(size), size) * quant[c.tq][zz];
									k++;
								}
								size_t px = ((size_t)mx * c.h + (size_t)bx) * 8, py = ((size_t)my * c.v + (size_t)by) * 8;
								idct_block(coef, &c.plane[py * (size_t)c.blocksW * 8 + px], c.blocksW * 8);
							}
						}
					}
					if (restartInterval) untilRestart--;
				}
			}
			// assemble: chroma samples are replicated, YCbCr -> RGB as in JFIF
			out.assign((size_t)W * (size_t)H, 0);
			for (int y = 0; y < H; y++) {
				for (int x = 0; x < W; x++) {
					auto sample = [&](const Component& c) {
						int sx = x * c.h / hmax, sy = y * c.v / vmax;
						return (int)c.plane[(size_t)sy * (size_t)c.blocksW * 8 + (size_t)sx];
					};
					int R, G, B;
					if (comps.size() == 1) { R = G = B = sample(comps[0]); }
					else {
						int Y = sample(comps[0]), cb = sample(comps[1]) - 128, cr = sample(comps[2]) - 128;
						// 16-bit fixed point of 1.402, 0.344136, 0.714136, 1.772
						R = Y + ((91881 * cr + 32768) >> 16);
						G = Y + ((-22554 * cb - 46802 * cr + 32768) >> 16);
						B = Y + ((116130 * cb + 32768) >> 16);
					}
					out[(size_t)y * (size_t)W + (size_t)x] = (uint32_t)clamp8(B) | ((uint32_t)clamp8(G) << 8) | ((uint32_t)clamp8(R) << 16) | (255u << 24);
				}
			}
			width = W; height = H;
			return true;
		}
		pos = segEnd;
	}
	return false;
}

bool LoadJpeg(const char* path, std::vector<uint32_t>& out, int32_t& width, int32_t& height)
{
	try { return LoadJpegUnguarded(path, out, width, height); }
	catch (...) { out.clear(); return false; }      // allocation failure on a hostile size: a decode failure like any other
}

bool LoadImageFile(const char* path, std::vector<uint32_t>& out, int32_t& width, int32_t& height)
{
	FILE* f = fopen(path, "rb");
	if (!f) return false;
	uint8_t magic[4] = {};
	size_t got = fread(magic, 1, 4, f);
	fclose(f);
	if (got < 4) return false;
	if (magic[0] == 0xFF && magic[1] == 0xD8) return LoadJpeg(path, out, width, height);
	return LoadPng(path, out, width, height);
}

} // namespace pth
