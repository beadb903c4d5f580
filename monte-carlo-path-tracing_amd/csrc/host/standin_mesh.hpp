// Procedural stand-ins for mesh files a scene names but that are not on disk (standin_mesh.cpp).
#ifndef MCPT_HOST_STANDIN_MESH_HPP
#define MCPT_HOST_STANDIN_MESH_HPP

#include <map>
#include <string>
#include <vector>

#include "asset_io.hpp"

namespace mcpt
{

class StandinTable
{
public:
    StandinTable() = default;
    explicit StandinTable(const std::string &text); // throws on a malformed line
    bool Has(const std::string &name) const;        // name as written in the XML's filename attribute
    MeshData Build(const std::string &name) const;
    bool empty() const { return lines_.empty(); }

private:
    std::map<std::string, std::vector<std::string>> lines_; // the parts of a file's mesh, in table order
};

} // namespace mcpt

#endif // MCPT_HOST_STANDIN_MESH_HPP
