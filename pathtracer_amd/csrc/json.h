// json.h -- a small recursive-descent JSON reader for scene files.
//
// Values keep object members in file order (a vector of pairs) so that anything derived
// from iteration order is reproducible.  Numbers are kept as double.  Parse errors throw
// std::runtime_error with the byte offset.
#pragma once
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace ptx {
namespace json {

struct Value {
	enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
	bool b = false;
	double num = 0.0;
	std::string str;
	std::vector<Value> items;                              // Array
	std::vector<std::pair<std::string, Value>> members;    // Object, file order

	const Value* find(const std::string& key) const {
		if (kind != Object) return nullptr;
		for (const auto& m : members)
			if (m.first == key) return &m.second;
		return nullptr;
	}
	bool is_string() const { return kind == String; }
	bool is_number() const { return kind == Number; }
};

class Parser {
public:
	explicit Parser(const std::string& text) : s(text) {}
	Value parse() {
		Value v = value();
		skip();
		if (pos != s.size()) fail("trailing characters");
		return v;
	}

private:
	const std::string& s;
	size_t pos = 0;

	[[noreturn]] void fail(const char* what) const { throw std::runtime_error(std::string("JSON: ") + what + " at byte " + std::to_string(pos)); }
	void skip() {
		while (pos < s.size() && (s[pos] == ' ' || s[pos] == '\t' || s[pos] == '\n' || s[pos] == '\r')) pos++;
	}
	bool eat(char c) {
		skip();
		if (pos < s.size() && s[pos] == c) {
			pos++;
			return true;
		}
		return false;
	}
	void word(const char* w) {
		for (const char* p = w; *p; p++, pos++)
			if (pos >= s.size() || s[pos] != *p) fail("bad literal");
	}
	static void utf8(std::string& out, uint32_t cp) {
		if (cp < 0x80) out += char(cp);
		else if (cp < 0x800) out += char(0xC0 | (cp >> 6)), out += char(0x80 | (cp & 0x3F));
		else if (cp < 0x10000) out += char(0xE0 | (cp >> 12)), out += char(0x80 | ((cp >> 6) & 0x3F)), out += char(0x80 | (cp & 0x3F));
		else out += char(0xF0 | (cp >> 18)), out += char(0x80 | ((cp >> 12) & 0x3F)), out += char(0x80 | ((cp >> 6) & 0x3F)), out += char(0x80 | (cp & 0x3F));
	}
	uint32_t hex4() {
		if (pos + 4 > s.size()) fail("short \\u escape");
		uint32_t v = 0;
		for (int i = 0; i < 4; i++) {
			const char c = s[pos++];
			v <<= 4;
			if (c >= '0' && c <= '9') v |= uint32_t(c - '0');
			else if (c >= 'a' && c <= 'f') v |= uint32_t(c - 'a' + 10);
			else if (c >= 'A' && c <= 'F') v |= uint32_t(c - 'A' + 10);
			else fail("bad \\u escape");
		}
		return v;
	}
	std::string string() {
		if (!eat('"')) fail("expected string");
		std::string out;
		// long base64 blobs have no escapes: copy runs at once
		for (;;) {
			const size_t run = s.find_first_of("\"\\", pos);
			if (run == std::string::npos) fail("unterminated string");
			out.append(s, pos, run - pos);
			pos = run;
			if (s[pos++] == '"') return out;
			if (pos >= s.size()) fail("unterminated escape");
			const char e = s[pos++];
			switch (e) {
			case '"': out += '"'; break;
			case '\\': out += '\\'; break;
			case '/': out += '/'; break;
			case 'b': out += '\b'; break;
			case 'f': out += '\f'; break;
			case 'n': out += '\n'; break;
			case 'r': out += '\r'; break;
			case 't': out += '\t'; break;
			case 'u': {
				uint32_t cp = hex4();
				if (cp >= 0xD800 && cp < 0xDC00 && pos + 1 < s.size() && s[pos] == '\\' && s[pos + 1] == 'u') {
					pos += 2;
					const uint32_t lo = hex4();
					cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
				}
				utf8(out, cp);
				break;
			}
			default: fail("bad escape");
			}
		}
	}
	Value value() {
		skip();
		if (pos >= s.size()) fail("unexpected end");
		Value v;
		const char c = s[pos];
		if (c == '{') {
			pos++;
			v.kind = Value::Object;
			if (eat('}')) return v;
			do {
				skip();
				std::string key = string();
				if (!eat(':')) fail("expected ':'");
				v.members.emplace_back(std::move(key), value());
			} while (eat(','));
			if (!eat('}')) fail("expected '}'");
		} else if (c == '[') {
			pos++;
			v.kind = Value::Array;
			if (eat(']')) return v;
			do v.items.push_back(value());
			while (eat(','));
			if (!eat(']')) fail("expected ']'");
		} else if (c == '"') {
			v.kind = Value::String;
			v.str = string();
		} else if (c == 't') {
			word("true"), v.kind = Value::Bool, v.b = true;
		} else if (c == 'f') {
			word("false"), v.kind = Value::Bool, v.b = false;
		} else if (c == 'n') {
			word("null");
		} else {
			const char* begin = s.c_str() + pos;
			char* end = nullptr;
			v.num = std::strtod(begin, &end);
			if (end == begin) fail("unexpected character");
			pos += size_t(end - begin);
			v.kind = Value::Number;
		}
		return v;
	}
};

inline Value parse(const std::string& text) { return Parser(text).parse(); }

}  // namespace json
}  // namespace ptx
