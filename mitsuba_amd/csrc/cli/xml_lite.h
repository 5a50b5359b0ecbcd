// Minimal XML reader for Mitsuba scene files: elements, attributes, comments, processing instructions and the five
// predefined entities. Character data is ignored (the scene format keeps all values in attributes,
// src/librender/scenehandler.cpp:306-693).
#pragma once
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace xml_lite {

struct Element {
    std::string tag;
    std::map<std::string, std::string> attrs;
    std::vector<std::unique_ptr<Element>> children;
    int line = 0;

    bool has(const std::string &k) const { return attrs.count(k) != 0; }
    std::string get(const std::string &k, const std::string &def = "") const {
        auto it = attrs.find(k);
        return it == attrs.end() ? def : it->second;
    }
};

class Parser {
public:
    explicit Parser(const std::string &text) : s(text) {}

    std::unique_ptr<Element> parse() {
        skipMisc();
        std::unique_ptr<Element> root = element();
        skipMisc();
        if (pos != s.size()) fail("trailing content after the root element");
        return root;
    }

private:
    const std::string &s;
    size_t pos = 0;
    int line = 1;

    [[noreturn]] void fail(const std::string &msg) const {
        throw std::runtime_error("XML parse error at line " + std::to_string(line) + ": " + msg);
    }
    bool startsWith(const char *lit) const { return s.compare(pos, std::char_traits<char>::length(lit), lit) == 0; }
    void advance(size_t n) {
        for (size_t i = 0; i < n && pos < s.size(); ++i, ++pos)
            if (s[pos] == '\n') ++line;
    }
    void skipSpace() {
        while (pos < s.size() && (s[pos] == ' ' || s[pos] == '\t' || s[pos] == '\r' || s[pos] == '\n')) advance(1);
    }
    void skipUntil(const char *lit) {
        size_t e = s.find(lit, pos);
        if (e == std::string::npos) fail(std::string("unterminated construct, expected ") + lit);
        advance(e - pos + std::char_traits<char>::length(lit));
    }
    // whitespace, comments, processing instructions, doctype and character data
    void skipMisc() {
        while (pos < s.size()) {
            if (startsWith("<!--")) skipUntil("-->");
            else if (startsWith("<?")) skipUntil("?>");
            else if (startsWith("<!")) skipUntil(">");
            else if (s[pos] != '<') advance(1);
            else break;
        }
    }
    static bool nameChar(char c) {
        return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9') || c == '_' || c == '-' || c == ':' || c == '.';
    }
    std::string name() {
        size_t b = pos;
        while (pos < s.size() && nameChar(s[pos])) ++pos;
        if (b == pos) fail("expected a name");
        return s.substr(b, pos - b);
    }
    std::string decode(const std::string &v) const {
        std::string out;
        for (size_t i = 0; i < v.size(); ++i) {
            if (v[i] != '&') { out += v[i]; continue; }
            size_t e = v.find(';', i);
            if (e == std::string::npos) fail("unterminated entity");
            std::string ent = v.substr(i + 1, e - i - 1);
            if (ent == "lt") out += '<';
            else if (ent == "gt") out += '>';
            else if (ent == "amp") out += '&';
            else if (ent == "quot") out += '"';
            else if (ent == "apos") out += '\'';
            else if (!ent.empty() && ent[0] == '#') out += (char) (ent[1] == 'x' ? std::stoi(ent.substr(2), nullptr, 16) : std::stoi(ent.substr(1)));
            else fail("unknown entity &" + ent + ";");
            i = e;
        }
        return out;
    }
    std::unique_ptr<Element> element() {
        if (pos >= s.size() || s[pos] != '<') fail("expected '<'");
        std::unique_ptr<Element> el(new Element());
        el->line = line;
        advance(1);
        el->tag = name();
        while (true) {
            skipSpace();
            if (pos >= s.size()) fail("unterminated start tag");
            if (startsWith("/>")) { advance(2); return el; }
            if (s[pos] == '>') { advance(1); break; }
            std::string key = name();
            skipSpace();
            if (pos >= s.size() || s[pos] != '=') fail("expected '=' after attribute name");
            advance(1);
            skipSpace();
            if (pos >= s.size() || (s[pos] != '"' && s[pos] != '\'')) fail("expected a quoted attribute value");
            const char q = s[pos];
            advance(1);
            size_t e = s.find(q, pos);
            if (e == std::string::npos) fail("unterminated attribute value");
            el->attrs[key] = decode(s.substr(pos, e - pos));
            advance(e - pos + 1);
        }
        while (true) {
            skipMisc();
            if (pos >= s.size()) fail("missing end tag for <" + el->tag + ">");
            if (startsWith("</")) {
                advance(2);
                std::string endName = name();
                if (endName != el->tag) fail("end tag </" + endName + "> does not match <" + el->tag + ">");
                skipSpace();
                if (pos >= s.size() || s[pos] != '>') fail("expected '>'");
                advance(1);
                return el;
            }
            el->children.push_back(element());
        }
    }
};

}  // namespace xml_lite
