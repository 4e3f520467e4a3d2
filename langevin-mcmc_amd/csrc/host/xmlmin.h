// Tiny XML reader for the Mitsuba-0.5 scene subset the reference parses with pugixml
// (/root/reference/src/parsescene.cpp:592-625): elements, attributes, comments, the <?xml?>
// declaration.  No entities beyond the five predefined ones, no CDATA, no namespaces.
#pragma once
#include <cctype>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace lmc {

struct XmlNode {
    std::string name;
    std::vector<std::pair<std::string, std::string>> attrs;
    std::vector<std::unique_ptr<XmlNode>> children;
    bool has(const std::string &k) const {
        for (auto &a : attrs)
            if (a.first == k) return true;
        return false;
    }
    // like pugixml: a missing attribute reads as ""
    std::string attr(const std::string &k) const {
        for (auto &a : attrs)
            if (a.first == k) return a.second;
        return "";
    }
    const XmlNode *child(const std::string &n) const {
        for (auto &c : children)
            if (c->name == n) return c.get();
        return nullptr;
    }
};

class XmlParser {
   public:
    explicit XmlParser(const std::string &s) : s_(s) {}
    std::unique_ptr<XmlNode> parse() {
        std::unique_ptr<XmlNode> root(new XmlNode);
        root->name = "#document";
        while (true) {
            skipMisc();
            if (p_ >= s_.size()) break;
            root->children.push_back(element());
        }
        return root;
    }

   private:
    const std::string &s_;
    size_t p_ = 0;
    [[noreturn]] void fail(const char *m) { throw std::runtime_error(std::string("Parse error: ") + m + " at offset " + std::to_string(p_)); }
    void ws() {
        while (p_ < s_.size() && isspace((unsigned char)s_[p_])) p_++;
    }
    bool starts(const char *t) { return s_.compare(p_, strlen(t), t) == 0; }
    void skipMisc() {
        for (;;) {
            ws();
            if (starts("<!--")) {
                size_t e = s_.find("-->", p_);
                if (e == std::string::npos) fail("unterminated comment");
                p_ = e + 3;
            } else if (starts("<?")) {
                size_t e = s_.find("?>", p_);
                if (e == std::string::npos) fail("unterminated declaration");
                p_ = e + 2;
            } else if (starts("<!")) {
                size_t e = s_.find('>', p_);
                if (e == std::string::npos) fail("unterminated doctype");
                p_ = e + 1;
            } else if (p_ < s_.size() && s_[p_] != '<') {
                p_++;  // stray text between elements is ignored
            } else
                return;
        }
    }
    std::string ident() {
        size_t b = p_;
        while (p_ < s_.size() && (isalnum((unsigned char)s_[p_]) || s_[p_] == '_' || s_[p_] == '-' || s_[p_] == ':' || s_[p_] == '.')) p_++;
        if (b == p_) fail("expected a name");
        return s_.substr(b, p_ - b);
    }
    static std::string unescape(const std::string &v) {
        std::string o;
        for (size_t i = 0; i < v.size(); i++) {
            if (v[i] != '&') {
                o.push_back(v[i]);
                continue;
            }
            struct { const char *e; char c; } tab[] = {{"&lt;", '<'}, {"&gt;", '>'}, {"&amp;", '&'}, {"&quot;", '"'}, {"&apos;", '\''}};
            bool hit = false;
            for (auto &t : tab)
                if (v.compare(i, strlen(t.e), t.e) == 0) {
                    o.push_back(t.c);
                    i += strlen(t.e) - 1;
                    hit = true;
                    break;
                }
            if (!hit) o.push_back('&');
        }
        return o;
    }
    std::unique_ptr<XmlNode> element() {
        if (s_[p_] != '<') fail("expected '<'");
        p_++;
        std::unique_ptr<XmlNode> n(new XmlNode);
        n->name = ident();
        for (;;) {
            ws();
            if (p_ >= s_.size()) fail("unterminated tag");
            if (starts("/>")) {
                p_ += 2;
                return n;
            }
            if (s_[p_] == '>') {
                p_++;
                break;
            }
            std::string k = ident();
            ws();
            if (p_ >= s_.size() || s_[p_] != '=') fail("expected '='");
            p_++;
            ws();
            char q = s_[p_];
            if (q != '"' && q != '\'') fail("expected quoted attribute value");
            size_t e = s_.find(q, p_ + 1);
            if (e == std::string::npos) fail("unterminated attribute value");
            n->attrs.emplace_back(k, unescape(s_.substr(p_ + 1, e - p_ - 1)));
            p_ = e + 1;
        }
        for (;;) {
            skipMisc();
            if (p_ >= s_.size()) fail("unterminated element");
            if (starts("</")) {
                p_ += 2;
                std::string cn = ident();
                if (cn != n->name) fail("mismatched closing tag");
                ws();
                if (p_ >= s_.size() || s_[p_] != '>') fail("expected '>'");
                p_++;
                return n;
            }
            n->children.push_back(element());
        }
    }
};

}  // namespace lmc
