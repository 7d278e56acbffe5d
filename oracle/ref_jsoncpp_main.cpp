// ORACLE -- test infrastructure only.
// Driver around the reference's vendored external/jsoncpp.cpp + external/json/json.h (compiled from where they lie under
// /root/reference; no reference source is copied into this repo).  Parses each file exactly as ConfigJSON::CreateFromFile does
// (reference src/config.cpp:266-272: Json::Reader, parse(file, root, collectComments = false)) and prints the parsed tree in
// a canonical text form, one file per line:
//   <path>\t<tree>      tree := {"key":tree,...} (keys sorted) | [tree,...] | "string" | i<int> | u<uint> | r<real %.17g> | true|false|null
//   <path>\tERROR       when the reader rejects the file
#include <cstdio>
#include <fstream>
#include <string>
#include "json/json.h"

static void dump(const Json::Value& v, std::string& o) {
    char buf[64];
    switch (v.type()) {
    case Json::nullValue: o += "null"; break;
    case Json::booleanValue: o += v.asBool() ? "true" : "false"; break;
    case Json::intValue: snprintf(buf, sizeof(buf), "i%lld", (long long)v.asLargestInt()); o += buf; break;
    case Json::uintValue: snprintf(buf, sizeof(buf), "u%llu", (unsigned long long)v.asLargestUInt()); o += buf; break;
    case Json::realValue: snprintf(buf, sizeof(buf), "r%.17g", v.asDouble()); o += buf; break;
    case Json::stringValue: {
        o += '"';
        for (unsigned char c : v.asString()) { if (c == '"' || c == '\\') { o += '\\'; o += (char)c; } else if (c < 0x20) { snprintf(buf, sizeof(buf), "\\u%04x", c); o += buf; } else o += (char)c; }
        o += '"';
        break;
    }
    case Json::arrayValue:
        o += '[';
        for (Json::ArrayIndex i = 0; i < v.size(); i++) { if (i) o += ','; dump(v[i], o); }
        o += ']';
        break;
    case Json::objectValue: {
        o += '{';
        bool first = true;
        for (const std::string& k : v.getMemberNames()) { // std::map order: sorted
            if (!first) o += ',';
            first = false;
            o += '"'; o += k; o += "\":";
            dump(v[k], o);
        }
        o += '}';
        break;
    }
    }
}

int main(int argc, char** argv) {
    for (int i = 1; i < argc; i++) {
        std::ifstream file(argv[i], std::ios::in);
        Json::Value root;
        Json::Reader reader;
        reader.parse(file, root, false);
        std::string o;
        if (!file || !reader.good()) o = "ERROR"; else dump(root, o);
        printf("%s\t%s\n", argv[i], o.c_str());
    }
    return 0;
}
