#pragma once
#include <deal.II/base/exceptions.h>
#include <deal.II/base/subscriptor.h>
#include <functional>
#include <memory>
#include <string>
#include <tuple>
#include <vector>
namespace dealii
{
  namespace Patterns
  {
    class PatternBase
    {
    public:
      virtual ~PatternBase() = default;
      virtual bool match(const std::string &) const { return true; }
      enum OutputStyle { Machine, Text, LaTeX };
      virtual std::string description(const OutputStyle = Machine) const { return {}; }
      virtual std::unique_ptr<PatternBase> clone() const { return nullptr; }
    };
    class Selection : public PatternBase
    {
    public:
      explicit Selection(const std::string &) {}
    };
    class Anything : public PatternBase {};
    class Bool : public PatternBase {};
    class Double : public PatternBase { public: Double(double = 0., double = 0.) {} };
    class Integer : public PatternBase { public: Integer(int = 0, int = 0) {} };
    class List : public PatternBase { public: List(const PatternBase &, unsigned int = 0, unsigned int = 0, const std::string & = ",") {} };
    namespace Tools
    {
      template <class T, class Enable = void>
      struct Convert {
        static std::unique_ptr<Patterns::PatternBase> to_pattern() { return std::make_unique<Patterns::Anything>(); }
        static std::string to_string(const T &, const Patterns::PatternBase & = *Convert<T>::to_pattern()) { return {}; }
        static T to_value(const std::string &, const Patterns::PatternBase & = *Convert<T>::to_pattern()) { return T(); }
      };
      struct ExcNoMatch : ExceptionBase { ExcNoMatch(const std::string &, const std::string &) {} };
    }
    using Tools::ExcNoMatch;
  }
  using Patterns::Tools::ExcNoMatch;

  class ParameterHandler
  {
  public:
    void enter_subsection(const std::string &);
    void leave_subsection();
    void declare_entry(const std::string &, const std::string &, const Patterns::PatternBase & = Patterns::Anything(), const std::string & = "");
    std::string get(const std::string &) const;
    double get_double(const std::string &) const;
    long get_integer(const std::string &) const;
    bool get_bool(const std::string &) const;
    template <class T> void add_parameter(const std::string &, T &, const std::string & = "", const Patterns::PatternBase & = *Patterns::Tools::Convert<T>::to_pattern());
    void add_action(const std::string &, const std::function<void(const std::string &)> &);
    enum OutputStyle { Text = 1, LaTeX = 2, Description = 4, XML = 8, JSON = 16, PRM = 32, Short = 64, KeepDeclarationOrder = 128, ShortPRM = 192 | 32 };
    std::ostream &print_parameters(std::ostream &, const unsigned int) const;
    void print_parameters(const std::string &, const unsigned int) const;
    void log_parameters(class LogStream &, const unsigned int = 0);
  };

  class ParameterAcceptor : public Subscriptor
  {
  public:
    explicit ParameterAcceptor(const std::string &section_name = "") : section(section_name) {}
    virtual ~ParameterAcceptor() = default;
    static void initialize(const std::string &filename = "", const std::string &output_filename = "",
                           const ParameterHandler::OutputStyle = ParameterHandler::Short, ParameterHandler &prm = ParameterAcceptor::prm,
                           const ParameterHandler::OutputStyle = ParameterHandler::Short);
    virtual void declare_parameters(ParameterHandler &) {}
    virtual void parse_parameters(ParameterHandler &) {}
    struct Signal {
      template <typename F> void connect(F &&) {}
      void operator()() const {}
    };
    Signal declare_parameters_call_back;
    Signal parse_parameters_call_back;
    std::string get_section_name() const { return section; }
    std::vector<std::string> get_section_path() const { return {}; }
    template <class ParameterType>
    void add_parameter(const std::string &, ParameterType &, const std::string & = "", ParameterHandler & = prm,
                       const Patterns::PatternBase & = *Patterns::Tools::Convert<ParameterType>::to_pattern()) {}
    void enter_subsection(const std::string &) {}
    void leave_subsection() {}
    void enter_my_subsection(ParameterHandler & = prm) {}
    void leave_my_subsection(ParameterHandler & = prm) {}
    static ParameterHandler prm;
  protected:
    const std::string section;
  };
}
