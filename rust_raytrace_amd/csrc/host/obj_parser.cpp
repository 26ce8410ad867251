// obj_parser.cpp — restatement of raytrace_lib/src/obj_parser.rs:20-73.
// "v x y z" and "f a[/..] b c ..." lines only, 1-based indices, first three
// corners of a face; everything else is ignored.  Where the reference panics
// (unwrap / assert / index out of bounds) this throws std::runtime_error.
#include <cerrno>
#include <cstdlib>
#include <fstream>
#include <sstream>

#include "raytrace.hpp"

namespace raytrace {
namespace obj_parser {

namespace {
// str::split_whitespace
std::vector<std::string> fields(const std::string& s) {
    std::vector<std::string> out;
    size_t i = 0;
    while (i < s.size()) {
        while (i < s.size() && isspace((unsigned char)s[i])) i++;
        size_t j = i;
        while (j < s.size() && !isspace((unsigned char)s[j])) j++;
        if (j > i) out.push_back(s.substr(i, j - i));
        i = j;
    }
    return out;
}
float parse_f32(const std::string& tok, const std::string& line) {
    char* end = nullptr;
    errno = 0;
    const float x = strtof(tok.c_str(), &end);  // correctly rounded decimal -> f32, like Rust's parse::<f32>
    if (end == tok.c_str() || *end != '\0') throw std::runtime_error("parse_obj: bad number in vertex line: " + line);
    return x;
}
size_t parse_index(const std::string& tok, const std::string& line) {
    const std::string head = tok.substr(0, tok.find('/'));  // x.split('/').nth(0)
    if (head.empty() || head.find_first_not_of("0123456789") != std::string::npos)
        throw std::runtime_error("parse_obj: bad index in face line: " + line);
    return (size_t)strtoull(head.c_str(), nullptr, 10);
}
}  // namespace

std::vector<Triangle> parse_obj(const std::string& path, const Vec3& offset, float scale,
                                const std::tuple<Vec3, Vec3, Vec3>& transform, const SurfaceKind& surface,
                                float edge_thickness) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("parse_obj: cannot read " + path);
    std::vector<Vec3> vertices;
    std::vector<std::vector<size_t>> faces;
    std::string line;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.compare(0, 2, "v ") == 0) {
            const auto parts = fields(line.substr(2));
            if (parts.size() != 3) throw std::runtime_error("parse_obj: vertex line needs exactly 3 numbers: " + line);
            vertices.push_back(make_vec(parse_f32(parts[0], line), parse_f32(parts[1], line), parse_f32(parts[2], line)));
        } else if (line.compare(0, 2, "f ") == 0) {
            std::vector<size_t> corners;
            for (const auto& tok : fields(line.substr(2))) corners.push_back(parse_index(tok, line));
            faces.push_back(std::move(corners));
        }
    }
    std::vector<Triangle> objs;
    objs.reserve(faces.size());
    for (const auto& face : faces) {
        if (face.size() < 3) throw std::runtime_error("parse_obj: face with fewer than 3 corners");
        Vec3 pts[3];
        for (int k = 0; k < 3; k++) {
            if (face[k] < 1 || face[k] > vertices.size()) throw std::runtime_error("parse_obj: face index out of range");
            pts[k] = vertices[face[k] - 1].mult(scale).change_basis(transform).add(offset);
        }
        objs.push_back(make_triangle(pts, surface, edge_thickness));
    }
    return objs;
}

}  // namespace obj_parser
}  // namespace raytrace
