// TEST SCAFFOLDING — the few jsoncpp names the reference's trajopt/problem_description.hpp mentions (Json::Value as a parameter
// type of the fromJson members).  Used by tests/test_adapters_compile.py only.
#pragma once
#include <string>
namespace Json
{
class Value
{
public:
  bool isMember(const std::string& key) const;
  const Value& operator[](const std::string& key) const;
};
}  // namespace Json
