// tools/dx_walk_stats.cpp -- analysis tool (not part of the library): what the lanes of k_dec_tiles do on a real sample.
// Decodes every coded band of a sample serially (the true code word sequence), cuts the payload into the 64-bit pieces of the chunk index
// and replays the tile kernel's walk piece by piece with the product's own multi-symbol table: steps per piece, steps that leave the table
// (long code words), and -- per output tile and per round of 64 pieces -- how many lanes have a piece and how long the round's slowest lane walks.
//   g++ -O2 -std=c++17 -shared -fPIC -Icineform-sdk_amd/csrc -Iinclude -Itests/hipemu tools/dx_walk_stats.cpp cineform-sdk_amd/csrc/cfhd_tables.cpp
//       cineform-sdk_amd/csrc/cfhd_bitstream.cpp cineform-sdk_amd/csrc/cfhd_metadata.cpp -o tools/_build/libdx_walk_stats.so      (driver: tools/dx_walk_stats.py)
#include "hip_emu.h"
dim3 threadIdx, blockIdx, blockDim, gridDim;
#include "cfhd_kernels.h"
#include "cfhd_bitstream.h"
#include "cfhd_entropy_jobs.h"
#include <map>
#include <vector>

using namespace cfhd;

namespace {
struct Word { uint32_t bit; uint32_t pos; int len; int kind; };     // a code word of the true sequence: first bit, raster position in front of it, bits incl. sign, kind
struct Trie { std::map<std::pair<int, uint32_t>, RawCode> m; int maxlen = 0; };

inline uint32_t bits_at(const uint8_t *p, size_t nbytes, uint64_t bit, int n)     // n <= 32 bits from bit position `bit`, MSB first; zeros behind the payload
{
	uint64_t v = 0;
	const uint64_t byte = bit >> 3;
	for (int k = 0; k < 8; k++) v = (v << 8) | (byte + k < nbytes ? p[byte + k] : 0);
	return (uint32_t)((v << (bit & 7)) >> (64 - n));
}
}

// out: per level group g (0: level 1 bands, 1: levels 2 and 3) 16 counters:
//  0 tiles, 1 rounds, 2 lanes with a piece inside (sum over rounds), 3 lane steps (sum), 4 wave steps (sum over rounds of the slowest lane), 5 wave steps in which some lane leaves the table,
//  6 lane steps that leave the table, 7 pieces (distinct), 8 coefficients, 9 payload bits, 10 wave steps if 64 consecutive pieces of a band shared a round (no tile boundaries),
//  11 long wave steps in that arrangement, 12 nonzero coefficients, 13 wave steps with long code words taken out of the lock step loop (max of short steps per round), 14 long rounds (max long per lane, summed), 15 rounds in (10)
extern "C" int dx_walk_stats(const uint8_t *sample, size_t size, int tile, uint64_t *out, uint64_t *hist /* 64: steps per piece */, uint64_t *lens /* [group][kind 0..2][32]: code words by length incl. sign */)
{
	ParsedSample ps;
	if (parse_sample(sample, size, &ps) != 0) return -1;
	static dev::DecIdxTables T; static bool ready = false;
	if (!ready) { if (!build_dec_index_tables(1, &T)) return -2; ready = true; }
	static Trie trie;
	if (trie.m.empty()) {
		RawCode codes[300];
		const int n = raw_codes(1, codes);
		for (int i = 0; i < n; i++) { trie.m[{ codes[i].len, codes[i].bits }] = codes[i]; if (codes[i].len > trie.maxlen) trie.maxlen = codes[i].len; }
	}
	for (int c = 0; c < ps.num_channels; c++)
		for (int lv = 0; lv < kNumLevels; lv++)
			for (int b = 1; b < 4; b++) {
				const ParsedBand &pb = ps.high[c][lv][b];
				if (!pb.present || !pb.bytes) continue;
				const int g = lv == 0 ? 0 : 1;
				uint64_t *o = out + 16 * g;
				const uint8_t *p = sample + pb.offset;
				const size_t nb = pb.bytes;
				// the true sequence
				std::vector<Word> words;
				uint64_t bit = 0; uint32_t pos = 0; uint64_t nonzero = 0;
				for (;;) {
					if (bit >= nb * 8) return -3;
					bool found = false;
					for (int len = 1; len <= trie.maxlen && !found; len++) {
						auto it = trie.m.find({ len, bits_at(p, nb, bit, len) });
						if (it == trie.m.end()) continue;
						found = true;
						const RawCode &rc = it->second;
						if (rc.kind == 2) { words.push_back(Word{ (uint32_t)bit, pos, len, 2 }); bit += len; goto done; }
						const int total = len + (rc.kind == 1 ? 1 : 0);
						words.push_back(Word{ (uint32_t)bit, pos, total, rc.kind });
						lens[(g * 3 + rc.kind) * 32 + total]++;
						pos += rc.kind == 1 ? 1u : (uint32_t)rc.payload; nonzero += rc.kind == 1;
						bit += total;
					}
					if (!found) return -4;
				}
			done:
				o[8] += pos; o[9] += bit; o[12] += nonzero;
				// pieces: the first code word that starts in each 64-bit piece
				const uint32_t npieces = (uint32_t)((bit + 63) / 64);
				std::vector<int> first(npieces, -1);
				for (size_t i = 0; i < words.size(); i++) { const uint32_t q = words[i].bit / 64; if (first[q] < 0) first[q] = (int)i; }
				// the tile kernel's walk of a piece: steps, long steps
				std::vector<uint8_t> steps(npieces, 0), longs(npieces, 0);
				std::vector<uint64_t> longmask(npieces, 0);       // bit s: step s of the piece leaves the table
				for (uint32_t q = 0; q < npieces; q++) {
					if (first[q] < 0) continue;
					size_t i = (size_t)first[q];
					uint32_t at = words[i].bit;
					int ns = 0, nl = 0;
					while (at < (q + 1) * 64u && i < words.size()) {
						const uint32_t win = bits_at(p, nb, at, dev::DX_KM);
						const uint32_t adv = T.multi[win].x & 15u;
						if (adv == 0) {
							if (words[i].kind == 2) { ns++; nl++; longmask[q] |= 1ull << (ns - 1); break; }
							at += words[i].len; i++; nl++; longmask[q] |= 1ull << ns;
						} else {
							uint32_t a = 0;
							while (a < adv) { a += words[i].len; i++; }
							if (a != adv) return -5;
							at += adv;
						}
						ns++;
						if (ns >= 63) break;
					}
					steps[q] = (uint8_t)ns; longs[q] = (uint8_t)nl;
					hist[ns < 63 ? ns : 63]++;
					o[7]++;
				}
				// tiles
				const uint32_t ncoef = (uint32_t)pb.height * (uint32_t)((pb.width + 7) / 8 * 8);
				uint32_t q0 = 0;
				for (uint32_t T0 = 0; T0 < ncoef; T0 += (uint32_t)tile) {
					const uint32_t T1 = T0 + (uint32_t)tile < ncoef ? T0 + (uint32_t)tile : ncoef;
					if (T0 >= pos) break;                            // behind the last code word: an empty tile (cleared, written; no pieces)
					// first piece: the last one whose first code word lies at or in front of T0
					while (q0 + 1 < npieces) {
						uint32_t nx = q0 + 1;
						while (nx < npieces && first[nx] < 0) nx++;
						if (nx < npieces && words[(size_t)first[nx]].pos <= T0) q0 = nx; else break;
					}
					o[0]++;
					uint32_t q = q0;
					bool over = false;
					while (!over && q < npieces) {
						int lanes = 0, mx = 0, mxshort = 0, mxlong = 0; uint64_t sum = 0, lsum = 0, lm = 0;
						for (uint32_t l = 0; l < 64 && q + l < npieces; l++) {
							const uint32_t qq = q + l;
							if (first[qq] < 0) continue;
							if (words[(size_t)first[qq]].pos >= T1) { over = true; continue; }
							lanes++; sum += steps[qq]; lsum += longs[qq]; lm |= longmask[qq];
							if (steps[qq] > mx) mx = steps[qq];
							if (steps[qq] - longs[qq] > mxshort) mxshort = steps[qq] - longs[qq];
							if (longs[qq] > mxlong) mxlong = longs[qq];
						}
						o[1]++; o[2] += lanes; o[3] += sum; o[4] += mx; o[5] += __builtin_popcountll(lm); o[6] += lsum; o[13] += mxshort; o[14] += mxlong;
						q += 64;
					}
				}
				// the same pieces, 64 consecutive ones per round whatever tile they belong to
				for (uint32_t q = 0; q < npieces; q += 64) {
					int mx = 0; uint64_t lm = 0;
					for (uint32_t l = 0; l < 64 && q + l < npieces; l++) { if (steps[q + l] > mx) mx = steps[q + l]; lm |= longmask[q + l]; }
					o[10] += mx; o[11] += __builtin_popcountll(lm); o[15]++;
					// two phases: short steps in lock step until every lane is done or waits at a long code word, then one long step for those, and again
					int at[64] = { 0 }; bool any = true;
					while (any) {
						int run = 0; any = false; bool lng = false;
						for (uint32_t l = 0; l < 64 && q + l < npieces; l++) {
							int n = 0;
							while (at[l] < steps[q + l] && !((longmask[q + l] >> at[l]) & 1ull)) { at[l]++; n++; }
							if (n > run) run = n;
							if (at[l] < steps[q + l]) { lng = true; at[l]++; }
							if (at[l] < steps[q + l]) any = true;
						}
						out[32] += run; out[33] += lng; out[34]++;
					}
				}
			}
	return 0;
}
