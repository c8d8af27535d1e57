// host_math.h -- the few pieces of lib/mat4.h and lib/bbox.h the scene flattener needs, with
// the reference's operation order so instance transforms come out bit for bit the same.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace ptx {

// 4x4 column-major matrix: m[col*4 + row] (lib/mat4.h).
struct Mat4f {
	float m[16];
	float at(int c, int r) const { return m[c * 4 + r]; }
};

inline Mat4f mat4_identity() {
	Mat4f r{};
	r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f;
	return r;
}

inline bool mat4_is_identity(const Mat4f& a) {  // `T != Mat4::I`, mat4.h:276-280
	for (int c = 0; c < 4; c++)
		for (int r = 0; r < 4; r++)
			if (a.at(c, r) != (c == r ? 1.0f : 0.0f)) return false;
	return true;
}

inline Mat4f mat4_transpose(const Mat4f& a) {
	Mat4f r;
	for (int c = 0; c < 4; c++)
		for (int k = 0; k < 4; k++) r.m[c * 4 + k] = a.m[k * 4 + c];
	return r;
}

// Mat4::inverse (mat4.h:325-377) and Mat4::det (mat4.h:237-262) are long sums of signed
// products whose grouping decides the last bits of the result.  They are kept here as data:
// each term is a sign followed by (col,row) digit pairs, multiplied left to right, and the
// terms are added left to right, which is how the C++ expressions in the reference parse.
// Adjugate entries: out[c][r] = sum of six signed products m[a][b]*m[c][d]*m[e][f], evaluated left to right.
static const char* const kAdjugate[16] = {
	"+122331 -132231 +132132 -112332 -122133 +112233",  // [0][0]
	"+032231 -022331 -032132 +012332 +022133 -012233",  // [0][1]
	"+021331 -031231 +031132 -011332 -021133 +011233",  // [0][2]
	"+031221 -021321 -031122 +011322 +021123 -011223",  // [0][3]
	"+132230 -122330 -132032 +102332 +122033 -102233",  // [1][0]
	"+022330 -032230 +032032 -002332 -022033 +002233",  // [1][1]
	"+031230 -021330 -031032 +001332 +021033 -001233",  // [1][2]
	"+021320 -031220 +031022 -001322 -021023 +001223",  // [1][3]
	"+112330 -132130 +132031 -102331 -112033 +102133",  // [2][0]
	"+032130 -012330 -032031 +002331 +012033 -002133",  // [2][1]
	"+011330 -031130 +031031 -001331 -011033 +001133",  // [2][2]
	"+031120 -011320 -031021 +001321 +011023 -001123",  // [2][3]
	"+122130 -112230 -122031 +102231 +112032 -102132",  // [3][0]
	"+012230 -022130 +022031 -002231 -012032 +002132",  // [3][1]
	"+021130 -011230 -021031 +001231 +011032 -001132",  // [3][2]
	"+011220 -021120 +021021 -001221 -011022 +001122",  // [3][3]
};
// Determinant: 24 signed products of four entries, evaluated left to right.
static const char* const kDeterminant =
	"+03122130 -02132130 -03112230 +01132230 +02112330 -01122330 "
	"-03122031 +02132031 +03102231 -00132231 -02102331 +00122331 "
	"+03112032 -01132032 -03102132 +00132132 +01102332 -00112332 "
	"-02112033 +01122033 +02102133 -00122133 -01102233 +00112233";

inline float eval_signed_products(const Mat4f& a, const char* s) {
	float sum = 0.0f;
	bool first = true;
	while (*s) {
		while (*s == ' ') ++s;
		if (!*s) break;
		const bool neg = (*s == '-');
		++s;
		float prod = 0.0f;
		bool first_factor = true;
		while (*s >= '0' && *s <= '3') {
			const float f = a.at(s[0] - '0', s[1] - '0');
			prod = first_factor ? f : prod * f;
			first_factor = false;
			s += 2;
		}
		if (first) sum = neg ? -prod : prod;   // a leading term is never negated in the reference
		else sum = neg ? sum - prod : sum + prod;
		first = false;
	}
	return sum;
}

inline float mat4_det(const Mat4f& a) { return eval_signed_products(a, kDeterminant); }

inline Mat4f mat4_inverse(const Mat4f& a) {
	Mat4f r;
	for (int i = 0; i < 16; i++) r.m[i] = eval_signed_products(a, kAdjugate[i]);
	const float d = mat4_det(a);
	for (int i = 0; i < 16; i++) r.m[i] = r.m[i] / d;   // r /= m.det(), mat4.h:375
	return r;
}

// BBox::transform (lib/bbox.h:62-79) on a {min xyz, max xyz} box.
inline void box_transform(const Mat4f& t, float box[6]) {
	const float amin[3] = {box[0], box[1], box[2]}, amax[3] = {box[3], box[4], box[5]};
	for (int i = 0; i < 3; i++) box[i] = box[3 + i] = t.at(3, i);
	for (int i = 0; i < 3; i++)
		for (int j = 0; j < 3; j++) {
			const float a = t.at(j, i) * amin[j], b = t.at(j, i) * amax[j];
			if (a < b) box[i] += a, box[3 + i] += b;
			else box[i] += b, box[3 + i] += a;
		}
}

}  // namespace ptx
