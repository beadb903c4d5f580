// A small XML reader: elements, attributes, comments, declarations, CDATA and
// the five predefined entities — what Mitsuba-style scene files use.  Text
// content is ignored.  Attribute accessors follow the conversion rules the
// reference relies on through pugixml (`as_float` = strtod, `as_int` = strtol,
// `as_bool` = first character in "1tTyY").
#ifndef MCPT_HOST_XML_DOM_HPP
#define MCPT_HOST_XML_DOM_HPP

#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace mcpt
{
namespace xml
{

struct Node
{
    std::string name;
    std::vector<std::pair<std::string, std::string>> attrs;
    std::vector<std::unique_ptr<Node>> children;

    const std::string *Attr(const std::string &key) const
    {
        for (const auto &kv : attrs)
            if (kv.first == key)
                return &kv.second;
        return nullptr;
    }
    bool Has(const std::string &key) const { return Attr(key) != nullptr; }
    std::string Str(const std::string &key, const std::string &fallback = "") const
    {
        const std::string *v = Attr(key);
        return v ? *v : fallback;
    }
    float Float(const std::string &key, float fallback) const
    {
        const std::string *v = Attr(key);
        return v ? static_cast<float>(std::strtod(v->c_str(), nullptr)) : fallback;
    }
    int Int(const std::string &key, int fallback) const
    {
        const std::string *v = Attr(key);
        return v ? static_cast<int>(std::strtol(v->c_str(), nullptr, 10)) : fallback;
    }
    bool Bool(const std::string &key, bool fallback) const
    {
        const std::string *v = Attr(key);
        if (!v || v->empty())
            return fallback;
        const char c = (*v)[0];
        return c == '1' || c == 't' || c == 'T' || c == 'y' || c == 'Y';
    }
    // first child element with this tag name, or nullptr
    const Node *Child(const std::string &tag) const
    {
        for (const auto &c : children)
            if (c->name == tag)
                return c.get();
        return nullptr;
    }
};

class Parser
{
public:
    explicit Parser(const std::string &text) : s_(text) {}

    std::unique_ptr<Node> ParseDocument()
    {
        std::unique_ptr<Node> root;
        for (;;)
        {
            SkipMisc();
            if (pos_ >= s_.size())
                break;
            if (s_[pos_] != '<')
                Fail("text outside of the root element");
            std::unique_ptr<Node> e = ParseElement();
            if (!root)
                root = std::move(e);
        }
        if (!root)
            Fail("no root element");
        return root;
    }

private:
    [[noreturn]] void Fail(const std::string &what) const
    {
        size_t line = 1;
        for (size_t i = 0; i < pos_ && i < s_.size(); ++i)
            line += s_[i] == '\n';
        throw std::runtime_error("XML parse error at line " + std::to_string(line) + ": " + what);
    }
    bool StartsWith(const char *lit) const { return s_.compare(pos_, std::char_traits<char>::length(lit), lit) == 0; }
    void SkipSpace()
    {
        while (pos_ < s_.size() && (s_[pos_] == ' ' || s_[pos_] == '\t' || s_[pos_] == '\n' || s_[pos_] == '\r'))
            ++pos_;
    }
    void SkipUntil(const char *lit)
    {
        const size_t at = s_.find(lit, pos_);
        if (at == std::string::npos)
            Fail(std::string("unterminated construct, expected '") + lit + "'");
        pos_ = at + std::char_traits<char>::length(lit);
    }
    // whitespace, comments, processing instructions, doctype, stray text
    void SkipMisc()
    {
        for (;;)
        {
            SkipSpace();
            if (StartsWith("<!--"))
                SkipUntil("-->");
            else if (StartsWith("<?"))
                SkipUntil("?>");
            else if (StartsWith("<![CDATA["))
                SkipUntil("]]>");
            else if (StartsWith("<!"))
                SkipUntil(">");
            else
                return;
        }
    }
    std::string ParseName()
    {
        const size_t start = pos_;
        while (pos_ < s_.size())
        {
            const char c = s_[pos_];
            if (c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '=' || c == '>' || c == '/')
                break;
            ++pos_;
        }
        if (pos_ == start)
            Fail("expected a name");
        return s_.substr(start, pos_ - start);
    }
    static std::string Unescape(const std::string &raw)
    {
        std::string out;
        out.reserve(raw.size());
        for (size_t i = 0; i < raw.size(); ++i)
        {
            if (raw[i] != '&')
            {
                out += raw[i];
                continue;
            }
            static const std::pair<const char *, char> table[] = {
                {"&amp;", '&'}, {"&lt;", '<'}, {"&gt;", '>'}, {"&quot;", '"'}, {"&apos;", '\''}};
            bool done = false;
            for (const auto &e : table)
            {
                const size_t n = std::char_traits<char>::length(e.first);
                if (raw.compare(i, n, e.first) == 0)
                {
                    out += e.second;
                    i += n - 1;
                    done = true;
                    break;
                }
            }
            if (!done)
                out += '&';
        }
        return out;
    }
    std::unique_ptr<Node> ParseElement()
    {
        ++pos_; // '<'
        std::unique_ptr<Node> node(new Node);
        node->name = ParseName();
        for (;;)
        {
            SkipSpace();
            if (pos_ >= s_.size())
                Fail("unterminated tag");
            if (s_[pos_] == '/')
            {
                if (pos_ + 1 >= s_.size() || s_[pos_ + 1] != '>')
                    Fail("malformed empty-element tag");
                pos_ += 2;
                return node;
            }
            if (s_[pos_] == '>')
            {
                ++pos_;
                break;
            }
            const std::string key = ParseName();
            SkipSpace();
            if (pos_ >= s_.size() || s_[pos_] != '=')
                Fail("attribute without value");
            ++pos_;
            SkipSpace();
            if (pos_ >= s_.size() || (s_[pos_] != '"' && s_[pos_] != '\''))
                Fail("attribute value must be quoted");
            const char quote = s_[pos_++];
            const size_t end = s_.find(quote, pos_);
            if (end == std::string::npos)
                Fail("unterminated attribute value");
            node->attrs.emplace_back(key, Unescape(s_.substr(pos_, end - pos_)));
            pos_ = end + 1;
        }
        // content
        for (;;)
        {
            // skip character data up to the next markup
            const size_t lt = s_.find('<', pos_);
            if (lt == std::string::npos)
                Fail("missing closing tag for <" + node->name + ">");
            pos_ = lt;
            if (StartsWith("<!--") || StartsWith("<?") || StartsWith("<!"))
            {
                SkipMisc();
                continue;
            }
            if (StartsWith("</"))
            {
                pos_ += 2;
                const std::string closing = ParseName();
                if (closing != node->name)
                    Fail("mismatched closing tag </" + closing + "> for <" + node->name + ">");
                SkipSpace();
                if (pos_ >= s_.size() || s_[pos_] != '>')
                    Fail("malformed closing tag");
                ++pos_;
                return node;
            }
            node->children.push_back(ParseElement());
        }
    }

    const std::string &s_;
    size_t pos_ = 0;
};

inline std::unique_ptr<Node> Parse(const std::string &text) { return Parser(text).ParseDocument(); }

} // namespace xml
} // namespace mcpt

#endif // MCPT_HOST_XML_DOM_HPP
