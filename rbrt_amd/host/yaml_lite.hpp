// yaml_lite.hpp — the subset of YAML the reference's scene files use (serde_yaml is not available):
// block maps and block lists by indentation, `- ` items that open a map on the same line, flow
// `[a, b]` / `{k: v}` collections, plain / single- / double-quoted scalars, `#` comments, `---`.
#pragma once
#include <string>
#include <utility>
#include <vector>

namespace yaml_lite {

struct Node {
    enum Kind { Null, Scalar, Map, List } kind = Null;
    std::string scalar;
    bool quoted = false;
    std::vector<std::pair<std::string, Node>> map;
    std::vector<Node> list;
    int line = 0;
    const Node* find(const std::string& key) const {
        for (auto& kv : map)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    bool is_null() const {
        return kind == Null || (kind == Scalar && !quoted && (scalar == "~" || scalar == "null" || scalar == "Null" ||
                                                              scalar == "NULL" || scalar.empty()));
    }
};

// Throws std::runtime_error with a line number on malformed input.
Node parse(const std::string& text);

}  // namespace yaml_lite
