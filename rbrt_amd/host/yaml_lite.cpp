#include "yaml_lite.hpp"

#include <stdexcept>

namespace yaml_lite {
namespace {

struct Line {
    int indent;
    std::string text;  // without indentation, comment and trailing blanks
    int no;
};

[[noreturn]] void bad(int line, const std::string& what) {
    throw std::runtime_error("yaml: line " + std::to_string(line) + ": " + what);
}

std::string rstrip(std::string s) {
    while (!s.empty() && (s.back() == ' ' || s.back() == '\t' || s.back() == '\r')) s.pop_back();
    return s;
}
std::string strip(const std::string& s) {
    size_t b = 0;
    while (b < s.size() && (s[b] == ' ' || s[b] == '\t')) ++b;
    return rstrip(s.substr(b));
}

// Removes a trailing comment: '#' at line start or preceded by a blank, outside quotes.
std::string strip_comment(const std::string& s) {
    char q = 0;
    for (size_t i = 0; i < s.size(); ++i) {
        char c = s[i];
        if (q) {
            if (c == q) q = 0;
        } else if (c == '"' || c == '\'') {
            q = c;
        } else if (c == '#' && (i == 0 || s[i - 1] == ' ' || s[i - 1] == '\t')) {
            return s.substr(0, i);
        }
    }
    return s;
}

std::vector<Line> split_lines(const std::string& text) {
    std::vector<Line> out;
    size_t pos = 0;
    int no = 0;
    while (pos <= text.size()) {
        size_t e = text.find('\n', pos);
        if (e == std::string::npos) e = text.size();
        std::string raw = text.substr(pos, e - pos);
        pos = e + 1;
        ++no;
        std::string s = rstrip(strip_comment(raw));
        size_t ind = 0;
        while (ind < s.size() && s[ind] == ' ') ++ind;
        if (ind < s.size() && s[ind] == '\t') bad(no, "tab used for indentation");
        std::string body = s.substr(ind);
        if (body.empty()) continue;
        if (ind == 0 && (body == "---" || body == "...")) continue;
        if (ind == 0 && body[0] == '%') continue;  // directive
        out.push_back(Line{int(ind), body, no});
    }
    return out;
}

// ---- flow / scalar parsing --------------------------------------------------------------------
struct Flow {
    const std::string& s;
    size_t i;
    int line;
    void ws() {
        while (i < s.size() && (s[i] == ' ' || s[i] == '\t')) ++i;
    }
    Node scalar_until(const char* stops) {
        ws();
        Node n;
        n.kind = Node::Scalar;
        n.line = line;
        if (i < s.size() && (s[i] == '"' || s[i] == '\'')) {
            char q = s[i++];
            n.quoted = true;
            while (i < s.size() && s[i] != q) {
                if (q == '"' && s[i] == '\\' && i + 1 < s.size()) {
                    char c = s[++i];
                    n.scalar += (c == 'n' ? '\n' : c == 't' ? '\t' : c);
                    ++i;
                } else {
                    n.scalar += s[i++];
                }
            }
            if (i >= s.size()) bad(line, "unterminated quoted string");
            ++i;
            return n;
        }
        size_t b = i;
        while (i < s.size()) {
            bool stop = false;
            for (const char* p = stops; *p; ++p)
                if (s[i] == *p) stop = true;
            if (stop) break;
            ++i;
        }
        n.scalar = strip(s.substr(b, i - b));
        return n;
    }
    Node value() {
        ws();
        if (i < s.size() && s[i] == '[') {
            ++i;
            Node n;
            n.kind = Node::List;
            n.line = line;
            ws();
            if (i < s.size() && s[i] == ']') {
                ++i;
                return n;
            }
            for (;;) {
                n.list.push_back(value());
                ws();
                if (i < s.size() && s[i] == ',') {
                    ++i;
                    continue;
                }
                if (i < s.size() && s[i] == ']') {
                    ++i;
                    return n;
                }
                bad(line, "expected ',' or ']' in flow sequence");
            }
        }
        if (i < s.size() && s[i] == '{') {
            ++i;
            Node n;
            n.kind = Node::Map;
            n.line = line;
            ws();
            if (i < s.size() && s[i] == '}') {
                ++i;
                return n;
            }
            for (;;) {
                Node k = scalar_until(":,}");
                ws();
                if (i >= s.size() || s[i] != ':') bad(line, "expected ':' in flow mapping");
                ++i;
                Node v = value();
                n.map.emplace_back(k.scalar, v);
                ws();
                if (i < s.size() && s[i] == ',') {
                    ++i;
                    continue;
                }
                if (i < s.size() && s[i] == '}') {
                    ++i;
                    return n;
                }
                bad(line, "expected ',' or '}' in flow mapping");
            }
        }
        return scalar_until(",]}");
    }
};

Node parse_inline(const std::string& text, int line) {
    std::string t = strip(text);
    if (!t.empty() && (t[0] == '[' || t[0] == '{')) {
        Flow f{t, 0, line};
        Node n = f.value();
        f.ws();
        if (f.i != t.size()) bad(line, "trailing characters after flow collection");
        return n;
    }
    Flow f{t, 0, line};
    if (!t.empty() && (t[0] == '"' || t[0] == '\'')) {
        Node n = f.scalar_until("");
        f.ws();
        if (f.i != t.size()) bad(line, "trailing characters after quoted scalar");
        return n;
    }
    Node n;
    n.kind = Node::Scalar;
    n.scalar = t;
    n.line = line;
    return n;
}

// Finds the ':' that ends a block-map key ("key: value" or "key:"); npos if the line is not a map entry.
size_t key_colon(const std::string& s) {
    char q = 0;
    for (size_t i = 0; i < s.size(); ++i) {
        char c = s[i];
        if (q) {
            if (c == q) q = 0;
        } else if ((c == '"' || c == '\'') && i == 0) {
            q = c;
        } else if (c == '[' || c == '{') {
            return std::string::npos;
        } else if (c == ':' && (i + 1 == s.size() || s[i + 1] == ' ')) {
            return i;
        }
    }
    return std::string::npos;
}

struct Parser {
    std::vector<Line> lines;
    size_t pos = 0;

    Node block(int indent) {
        if (pos >= lines.size()) return Node();
        const Line& first = lines[pos];
        if (first.text[0] == '-' && (first.text.size() == 1 || first.text[1] == ' ')) return list(indent);
        if (key_colon(first.text) != std::string::npos) return map(indent);
        Node n = parse_inline(first.text, first.no);
        ++pos;
        return n;
    }

    Node list(int indent) {
        Node n;
        n.kind = Node::List;
        n.line = lines[pos].no;
        while (pos < lines.size() && lines[pos].indent == indent && lines[pos].text[0] == '-' &&
               (lines[pos].text.size() == 1 || lines[pos].text[1] == ' ')) {
            Line& l = lines[pos];
            std::string rest = l.text.size() > 1 ? l.text.substr(1) : std::string();
            size_t skip = 0;
            while (skip < rest.size() && rest[skip] == ' ') ++skip;
            if (skip == rest.size()) {  // "-" alone: the item is the following deeper block
                ++pos;
                if (pos < lines.size() && lines[pos].indent > indent)
                    n.list.push_back(block(lines[pos].indent));
                else
                    n.list.push_back(Node());
            } else {
                // re-read the remainder of this line as the first line of a nested block
                l.indent = indent + 1 + int(skip);
                l.text = rest.substr(skip);
                n.list.push_back(block(l.indent));
            }
        }
        if (pos < lines.size() && lines[pos].indent > indent) bad(lines[pos].no, "unexpected indentation");
        return n;
    }

    Node map(int indent) {
        Node n;
        n.kind = Node::Map;
        n.line = lines[pos].no;
        while (pos < lines.size() && lines[pos].indent == indent) {
            const Line l = lines[pos];
            size_t c = key_colon(l.text);
            if (c == std::string::npos) bad(l.no, "expected 'key: value'");
            Node k = parse_inline(l.text.substr(0, c), l.no);
            std::string rest = strip(l.text.substr(c + 1));
            ++pos;
            Node v;
            if (!rest.empty()) {
                v = parse_inline(rest, l.no);
            } else if (pos < lines.size() && lines[pos].indent > indent) {
                v = block(lines[pos].indent);
            } else if (pos < lines.size() && lines[pos].indent == indent && lines[pos].text[0] == '-' &&
                       (lines[pos].text.size() == 1 || lines[pos].text[1] == ' ')) {
                v = list(indent);  // a sequence may sit at its parent key's indentation
            }
            v.line = v.line ? v.line : l.no;
            n.map.emplace_back(k.scalar, v);
        }
        if (pos < lines.size() && lines[pos].indent > indent) bad(lines[pos].no, "unexpected indentation");
        return n;
    }
};

}  // namespace

Node parse(const std::string& text) {
    Parser p;
    p.lines = split_lines(text);
    if (p.lines.empty()) return Node();
    Node n = p.block(p.lines[0].indent);
    if (p.pos != p.lines.size()) bad(p.lines[p.pos].no, "unexpected content (indentation does not match any open block)");
    return n;
}

}  // namespace yaml_lite
