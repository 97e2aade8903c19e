// ccsx_model_io.cpp — Arrow model parameter files and chemistry lookup.
//
// Reference behaviour restated (the trained tables themselves are not in the mount):
//   * docs/faq/chemistry.md:27-56  $SMRT_CHEMISTRY_BUNDLE_DIR/arrow/<model>.json injects consensus models; a chemistry
//     no model supports ends the run with "Unsupported chemistries found: (...)"
//   * docs/changelog.md:66          "Abort if chemistry information is missing in BAM header"
//   * docs/changelog.md:101         the used chemistry model is logged at INFO level
// The JSON schema is this library's own (the PacBio schema is not published in the mount): it carries exactly the
// ccsx_model blob plus the (BindingKit, SequencingKit, BasecallerVersion) triples the parameter set was trained for.
// Floats are written with %.9g, which round-trips binary32 exactly: file -> blob -> file is the identity.
#include "ccsx.h"
#include "ccsx_internal.h"

#include <dirent.h>

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace {

// ---- a minimal JSON reader: objects, arrays, strings (no \u escapes needed), numbers, true/false/null
struct JVal {
    enum Kind { NUL, NUM, STR, ARR, OBJ, BOOL } kind = NUL;
    double num = 0.0;
    std::string str;
    std::vector<JVal> arr;
    std::vector<std::pair<std::string, JVal>> obj;
    const JVal *get(const char *key) const
    {
        for (auto &kv : obj) if (kv.first == key) return &kv.second;
        return nullptr;
    }
};

struct JParser {
    const char *p, *end;
    std::string err;
    void ws() { while (p < end && std::isspace((unsigned char)*p)) ++p; }
    bool fail(const char *m) { if (err.empty()) err = m; return false; }
    bool parse_string(std::string &out)
    {
        if (p >= end || *p != '"') return fail("expected string");
        ++p; out.clear();
        while (p < end && *p != '"') {
            if (*p == '\\') {
                if (++p >= end) return fail("bad escape");
                switch (*p) { case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break; case 'b': out += '\b'; break;
                              case 'f': out += '\f'; break;
                              case 'u': {                  // \u00XX (what ccsx_model_to_json writes for control characters); no surrogates / non-ASCII
                                  unsigned v = 0;
                                  if (end - p < 5) return fail("bad \\u escape");
                                  for (int k = 1; k <= 4; ++k) { const int c = std::tolower((unsigned char)p[k]); if (!std::isxdigit(c)) return fail("bad \\u escape"); v = v * 16 + (unsigned)(c <= '9' ? c - '0' : c - 'a' + 10); }
                                  if (v == 0 || v > 0x7f) return fail("\\u escapes above U+007F are not supported");
                                  out += (char)v; p += 4; break; }
                              default: out += *p; }
                ++p;
            } else out += *p++;
        }
        if (p >= end) return fail("unterminated string");
        ++p;
        return true;
    }
    bool parse(JVal &v, int depth = 0)
    {
        if (depth > 16) return fail("nesting too deep");
        ws();
        if (p >= end) return fail("unexpected end");
        if (*p == '{') {
            v.kind = JVal::OBJ; ++p; ws();
            if (p < end && *p == '}') { ++p; return true; }
            for (;;) {
                ws();
                std::string k;
                if (!parse_string(k)) return false;
                ws();
                if (p >= end || *p != ':') return fail("expected ':'");
                ++p;
                v.obj.emplace_back(k, JVal());
                if (!parse(v.obj.back().second, depth + 1)) return false;
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; return true; }
                return fail("expected ',' or '}'");
            }
        }
        if (*p == '[') {
            v.kind = JVal::ARR; ++p; ws();
            if (p < end && *p == ']') { ++p; return true; }
            for (;;) {
                v.arr.emplace_back();
                if (!parse(v.arr.back(), depth + 1)) return false;
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == ']') { ++p; return true; }
                return fail("expected ',' or ']'");
            }
        }
        if (*p == '"') { v.kind = JVal::STR; return parse_string(v.str); }
        if (!std::strncmp(p, "true", std::min<size_t>(4, end - p)) && end - p >= 4) { v.kind = JVal::BOOL; v.num = 1; p += 4; return true; }
        if (!std::strncmp(p, "false", std::min<size_t>(5, end - p)) && end - p >= 5) { v.kind = JVal::BOOL; v.num = 0; p += 5; return true; }
        if (!std::strncmp(p, "null", std::min<size_t>(4, end - p)) && end - p >= 4) { v.kind = JVal::NUL; p += 4; return true; }
        char *q = nullptr;
        std::string tmp(p, std::min<size_t>(64, end - p));
        const double d = std::strtod(tmp.c_str(), &q);
        if (q == tmp.c_str()) return fail("unexpected character");
        v.kind = JVal::NUM; v.num = d; p += (q - tmp.c_str());
        return true;
    }
};

bool read_file(const std::string &path, std::string &out)
{
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    char buf[1 << 14];
    size_t n;
    out.clear();
    while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) { out.append(buf, n); if (out.size() > (16u << 20)) break; }
    std::fclose(f);
    return true;
}

bool fill_floats(const JVal *v, float *dst, const std::vector<int> &dims, size_t level, std::string &err, const char *what)
{
    if (!v || v->kind != JVal::ARR || (int)v->arr.size() != dims[level]) { err = std::string(what) + ": wrong shape"; return false; }
    size_t stride = 1;
    for (size_t k = level + 1; k < dims.size(); ++k) stride *= (size_t)dims[k];
    for (int i = 0; i < dims[level]; ++i) {
        if (level + 1 == dims.size()) {
            if (v->arr[i].kind != JVal::NUM) { err = std::string(what) + ": not a number"; return false; }
            dst[i] = (float)v->arr[i].num;
        } else if (!fill_floats(&v->arr[i], dst + (size_t)i * stride, dims, level + 1, err, what)) return false;
    }
    return true;
}

struct Triple { std::string bk, sk, bc; };

int parse_model(const std::string &text, ccsx_model *m, std::vector<Triple> *chems)
{
    JParser jp{text.data(), text.data() + text.size(), {}};
    JVal root;
    if (!jp.parse(root) || root.kind != JVal::OBJ) { ccsx_set_error("model json: " + (jp.err.empty() ? std::string("not an object") : jp.err)); return -1; }
    const JVal *ver = root.get("ConsensusModelVersion");
    if (!ver || ver->kind != JVal::STR || ver->str != "ccsx-1") { ccsx_set_error("model json: ConsensusModelVersion must be \"ccsx-1\""); return -1; }
    std::memset(m, 0, sizeof(*m));
    const JVal *nm = root.get("ChemistryName");
    if (!nm || nm->kind != JVal::STR || nm->str.empty() || nm->str.size() >= sizeof(m->name)) { ccsx_set_error("model json: ChemistryName missing or too long"); return -1; }
    std::strncpy(m->name, nm->str.c_str(), sizeof(m->name) - 1);
    std::string err;
    float snr[2];
    if (!fill_floats(root.get("SnrRange"), snr, {2}, 0, err, "SnrRange") ||
        !fill_floats(root.get("TransitionPolynomials"), &m->trans_poly[0][0][0], {CCSX_NCTX, 3, 4}, 0, err, "TransitionPolynomials") ||
        !fill_floats(root.get("EmissionMatch"), &m->em_match[0][0], {CCSX_NCTX, CCSX_NOBS}, 0, err, "EmissionMatch") ||
        !fill_floats(root.get("EmissionBranch"), &m->em_branch[0][0], {CCSX_NCTX, 3}, 0, err, "EmissionBranch") ||
        !fill_floats(root.get("EmissionStick"), &m->em_stick[0][0], {CCSX_NCTX, 3}, 0, err, "EmissionStick")) {
        ccsx_set_error("model json: " + err);
        return -1;
    }
    m->snr_lo = snr[0]; m->snr_hi = snr[1];
    if (!(m->snr_lo > 0.0f) || !(m->snr_hi >= m->snr_lo)) { ccsx_set_error("model json: SnrRange must be 0 < lo <= hi"); return -1; }
    if (!std::isfinite(m->snr_lo) || !std::isfinite(m->snr_hi)) { ccsx_set_error("model json: SnrRange must be finite"); return -1; }
    // every table entry is finite; emission tables are probabilities in (0, 1] (they go through log2 in the z-score parameters), their
    // rows sum to 1; polynomial coefficients are only required to be finite (the weights are clamped at 1e-6 on the device)
    for (int k = 0; k < CCSX_NCTX; ++k) {
        for (int mv = 0; mv < 3; ++mv) for (int c = 0; c < 4; ++c)
            if (!std::isfinite(m->trans_poly[k][mv][c])) { ccsx_set_error("model json: TransitionPolynomials must be finite"); return -1; }
        float s = 0.0f;
        for (int o = 0; o < CCSX_NOBS; ++o) {
            const float p = m->em_match[k][o];
            if (!std::isfinite(p) || !(p > 0.0f) || p > 1.0f) { ccsx_set_error("model json: EmissionMatch entries must be probabilities in (0, 1]"); return -1; }
            s += p;
        }
        if (s < 0.98f || s > 1.02f) { ccsx_set_error("model json: EmissionMatch rows must sum to 1"); return -1; }
        float sb = 0.0f, ss = 0.0f;
        for (int b = 0; b < 3; ++b) {
            const float pb = m->em_branch[k][b], ps = m->em_stick[k][b];
            if (!std::isfinite(pb) || !(pb > 0.0f) || pb > 1.0f || !std::isfinite(ps) || !(ps > 0.0f) || ps > 1.0f) {
                ccsx_set_error("model json: EmissionBranch / EmissionStick entries must be probabilities in (0, 1]"); return -1;
            }
            sb += pb; ss += ps;
        }
        if (sb < 0.98f || sb > 1.02f || ss < 0.98f || ss > 1.02f) { ccsx_set_error("model json: EmissionBranch / EmissionStick rows must sum to 1"); return -1; }
    }
    if (chems) {
        chems->clear();
        const JVal *cs = root.get("Chemistries");
        if (cs && cs->kind == JVal::ARR) for (const JVal &c : cs->arr) {
            const JVal *a = c.get("BindingKit"), *b = c.get("SequencingKit"), *d = c.get("BasecallerVersion");
            if (a && b && d && a->kind == JVal::STR && b->kind == JVal::STR && d->kind == JVal::STR) chems->push_back({a->str, b->str, d->str});
        }
    }
    return 0;
}

// basecaller versions are matched on major.minor ("5.0.0.6235" supports a model trained for "5.0")
std::string major_minor(const std::string &v)
{
    size_t d1 = v.find('.');
    if (d1 == std::string::npos) return v;
    size_t d2 = v.find('.', d1 + 1);
    return d2 == std::string::npos ? v : v.substr(0, d2);
}

bool triple_matches(const Triple &t, const char *bk, const char *sk, const char *bc)
{
    return t.bk == bk && t.sk == sk && major_minor(t.bc) == major_minor(bc);
}

// chemistries of the built-in parameter set SYN-1: the triple the synthetic subreads.bam writer stamps ([RECALL] Sequel II 2.0 part numbers)
const Triple kBuiltin[] = {{"101-789-500", "101-826-100", "5.0"}};

}  // namespace

extern "C" {

int ccsx_model_from_json(const char *json_text, ccsx_model *m)
{
    if (!json_text || !m) { ccsx_set_error("ccsx_model_from_json: null argument"); return -1; }
    return parse_model(json_text, m, nullptr);
}

int ccsx_model_load(const char *path, ccsx_model *m)
{
    if (!path || !m) { ccsx_set_error("ccsx_model_load: null argument"); return -1; }
    std::string text;
    if (!read_file(path, text)) { ccsx_set_error(std::string("ccsx_model_load: cannot read ") + path); return -1; }
    return parse_model(text, m, nullptr);
}

int64_t ccsx_model_to_json(const ccsx_model *m, const char *binding_kit, const char *sequencing_kit, const char *basecaller_version,
                           char *buf, int64_t cap)
{
    if (!m) { ccsx_set_error("ccsx_model_to_json: null argument"); return -1; }
    std::string s = "{\n  \"ConsensusModelVersion\": \"ccsx-1\",\n  \"ChemistryName\": \"";
    char name[sizeof(m->name) + 1] = {0};
    std::memcpy(name, m->name, sizeof(m->name));
    auto esc = [](const char *t) {                       // JSON string escaping (quotes, backslashes, control characters)
        std::string o;
        for (const unsigned char *p = (const unsigned char *)t; *p; ++p) {
            if (*p == '"' || *p == '\\') { o += '\\'; o += (char)*p; }
            else if (*p < 0x20) { char u[8]; std::snprintf(u, sizeof(u), "\\u%04x", (unsigned)*p); o += u; }
            else o += (char)*p;
        }
        return o;
    };
    s += esc(name);
    s += "\",\n  \"ModelForm\": \"PwSnr\",\n";
    char tmp[64];
    auto num = [&](float v) { std::snprintf(tmp, sizeof(tmp), "%.9g", (double)v); s += tmp; };
    s += "  \"Chemistries\": [";
    if (binding_kit && sequencing_kit && basecaller_version) {
        s += "{\"BindingKit\": \""; s += esc(binding_kit); s += "\", \"SequencingKit\": \""; s += esc(sequencing_kit);
        s += "\", \"BasecallerVersion\": \""; s += esc(basecaller_version); s += "\"}";
    }
    s += "],\n  \"SnrRange\": ["; num(m->snr_lo); s += ", "; num(m->snr_hi); s += "],\n";
    s += "  \"TransitionPolynomials\": [\n";
    for (int k = 0; k < CCSX_NCTX; ++k) {
        s += "    [";
        for (int mv = 0; mv < 3; ++mv) { s += "["; for (int c = 0; c < 4; ++c) { num(m->trans_poly[k][mv][c]); if (c < 3) s += ", "; } s += mv < 2 ? "], " : "]"; }
        s += k + 1 < CCSX_NCTX ? "],\n" : "]\n";
    }
    s += "  ],\n";
    auto mat = [&](const char *key, const float *p, int cols, bool last) {
        s += "  \""; s += key; s += "\": [\n";
        for (int k = 0; k < CCSX_NCTX; ++k) {
            s += "    [";
            for (int c = 0; c < cols; ++c) { num(p[k * cols + c]); if (c + 1 < cols) s += ", "; }
            s += k + 1 < CCSX_NCTX ? "],\n" : "]\n";
        }
        s += last ? "  ]\n" : "  ],\n";
    };
    mat("EmissionMatch", &m->em_match[0][0], CCSX_NOBS, false);
    mat("EmissionBranch", &m->em_branch[0][0], 3, false);
    mat("EmissionStick", &m->em_stick[0][0], 3, true);
    s += "}\n";
    if (buf && cap > 0) {
        const size_t n = std::min<size_t>(s.size(), (size_t)cap - 1);
        std::memcpy(buf, s.data(), n); buf[n] = 0;
    }
    return (int64_t)s.size();       // bytes needed (without the terminator): call with buf = NULL to size
}

// Model for a chemistry triple: $SMRT_CHEMISTRY_BUNDLE_DIR/arrow/*.json first (docs/faq/chemistry.md:27-56: injected models
// take precedence), then the built-in set.  -1 with "Unsupported chemistries found: (...)" when nothing supports it.
int ccsx_model_for_chemistry(const char *binding_kit, const char *sequencing_kit, const char *basecaller_version, ccsx_model *m)
{
    if (!binding_kit || !sequencing_kit || !basecaller_version || !m) { ccsx_set_error("ccsx_model_for_chemistry: null argument"); return -1; }
    if (const char *dir = std::getenv("SMRT_CHEMISTRY_BUNDLE_DIR")) {
        const std::string adir = std::string(dir) + "/arrow";
        std::vector<std::string> files;
        if (DIR *d = opendir(adir.c_str())) {
            while (dirent *e = readdir(d)) {
                const std::string n = e->d_name;
                if (n.size() > 5 && n.compare(n.size() - 5, 5, ".json") == 0) files.push_back(adir + "/" + n);
            }
            closedir(d);
        }
        std::sort(files.begin(), files.end());
        for (const std::string &f : files) {
            std::string text;
            ccsx_model cand;
            std::vector<Triple> chems;
            if (!read_file(f, text)) { std::fprintf(stderr, "ccsx: warning: cannot read %s\n", f.c_str()); continue; }
            if (parse_model(text, &cand, &chems)) {
                // a file that claims to be one of ours but does not validate is reported (an injected model with a typo must not
                // silently fall back to the built-in set: ADVICE r02); foreign json files in the bundle are skipped quietly
                if (text.find("\"ccsx-1\"") != std::string::npos) std::fprintf(stderr, "ccsx: warning: ignoring %s: %s\n", f.c_str(), ccsx_last_error());
                continue;
            }
            for (const Triple &t : chems) if (triple_matches(t, binding_kit, sequencing_kit, basecaller_version)) { *m = cand; return 0; }
        }
    }
    for (const Triple &t : kBuiltin) if (triple_matches(t, binding_kit, sequencing_kit, basecaller_version)) { ccsx_model_default(m); return 0; }
    ccsx_set_error(std::string("Unsupported chemistries found: (") + binding_kit + "/" + sequencing_kit + "/" + basecaller_version + ")");
    return -1;
}

}  // extern "C"
